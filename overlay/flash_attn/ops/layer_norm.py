from hyena_dna_amd.block import dropout_add_layer_norm  # noqa: F401
