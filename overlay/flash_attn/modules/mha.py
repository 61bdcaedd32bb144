from hyena_dna_amd.lm import MHA, ParallelMHA  # noqa: F401
