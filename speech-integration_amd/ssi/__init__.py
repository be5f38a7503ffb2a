"""MI355X-native drop-in for the training hot path of anilkeshwani/speech-integration.

Same public names as the reference's ``ssi`` package for that path (``ssi.trainer.Trainer``, ``ssi.loss.compute_loss``,
``ssi.model.setup_llama3_2_1b``, ``ssi.llama_configs``, ``ssi.train_utils``, ``ssi.eval``, ``ssi.optimizer``,
``ssi.lr_schedule``); underneath, hand-written HIP kernels for gfx950 reached through a C ABI (``include/ssi_hip.h``).
"""

__version__ = "0.1.0"
