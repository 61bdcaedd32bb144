from hyena_dna_amd.lm import Mlp, FusedMLP, ParallelFusedMLP  # noqa: F401
