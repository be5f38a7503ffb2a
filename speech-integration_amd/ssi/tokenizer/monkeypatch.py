"""Import path of the reference (``ssi.tokenizer.monkeypatch``, ``/root/reference/ssi/tokenizer/monkeypatch.py``)."""
from .llama3_pua import CL100K_PATTERN, CL100K_PATTERN_PUA, Llama3TokenizerPUA  # noqa: F401
