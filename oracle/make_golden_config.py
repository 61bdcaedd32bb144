#!/usr/bin/env python
"""Compose the reference's hg38 experiment config (configs/config.yaml + configs/experiment/hg38/hg38_hyena.yaml + the group
files their `defaults` lists name) with hyena_dna_amd.runner.compose_raw and store the composed, UNRESOLVED tree as
tests/golden/hg38_hyena_composed.json -- what `scripts/train_hg38.py` reads on a GPU box, where /root/reference does not exist.

    python oracle/make_golden_config.py          (build container only)

tests/test_runner.py checks that the live reference configs still compose to exactly this file."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("HYENA_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "hg38_hyena_composed.json")


def main():
    from hyena_dna_amd import runner
    cfg = runner.compose_raw(os.path.join(REF, "configs"), "hg38/hg38_hyena")
    doc = {"experiment": "hg38/hg38_hyena", "source": "composed from <reference>/configs by hyena_dna_amd.runner.compose_raw "
           "(oracle/make_golden_config.py); interpolations unresolved", "config": cfg}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
