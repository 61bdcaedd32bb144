from hyena_dna_amd.lm import GenerationMixin  # noqa: F401
