"""CPU oracle (test infrastructure only).  See oracle/llama_oracle.py for the import policy."""
