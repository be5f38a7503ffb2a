#!/usr/bin/env python
"""CPT entry point — same CLI as /root/reference/scripts/train_cpt.py:
    python scripts/train_cpt.py data=cpt/mls-hubert_large_ll60k-layer_22 [key=value ...]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

try:  # real Hydra if the environment has it, else the built-in composer with the same decorator shape
    import hydra
    main_decorator = hydra.main
except ImportError:
    from ssi.config import main as main_decorator

from ssi.train_utils import resolve_n_dsus
from ssi.trainer import Trainer


@main_decorator(config_path="../conf", config_name="cpt", version_base=None)
def main(cfg):
    resolve_n_dsus(cfg)
    trainer = Trainer(cfg)
    trainer.setup()
    trainer.train()
    trainer.cleanup()


if __name__ == "__main__":
    main()
