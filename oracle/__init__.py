"""CPU oracle (test infrastructure, never imported by ``maria_amd``).

See ``oracle/hotpath.py`` for the pinning statement: the restatement is pinned
by golden vectors only at the leaves the reference lets us import
(``maria.functions``, ``maria.beam``, ``maria.utils.linalg/rotations``);
for the jax float32 steps parity is UNPINNED.
"""
