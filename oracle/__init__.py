"""CPU oracle for the Dreamer hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (plain
PyTorch-CPU / numpy) of the reference algorithm for the hot path:
RSSM observe + imagine rollout, lambda-return, replay sequence gather.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``dreamer_amd``) never imports it.

Pinning: the reference ships no tests or golden vectors of its own.  The
oracle is pinned against outputs of the reference itself, captured in this
build's container by ``tests/golden/make_golden.py`` (which imports the
reference's Python modules from /root/reference) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks every function
here against those vectors.
"""
from .dreamer_oracle import *  # noqa: F401,F403
