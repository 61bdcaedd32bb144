from hyena_dna_amd.lm import sync_shared_params, all_gather_raw  # noqa: F401
