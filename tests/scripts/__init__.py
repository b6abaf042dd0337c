"""Measurement scripts that need the oracle (so they live under tests/, the only place allowed to import it) but are not
test modules: run them by hand, e.g. `python -m tests.scripts.encode_error` on a GPU box."""
