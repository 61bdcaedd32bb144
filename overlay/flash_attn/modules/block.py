from hyena_dna_amd.lm import Block  # noqa: F401
