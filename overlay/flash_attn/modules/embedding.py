from hyena_dna_amd.lm import GPT2Embeddings, ParallelGPT2Embeddings  # noqa: F401
