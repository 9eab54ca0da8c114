"""CPU oracle for the MAGPO Anakin hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy for the
integer / PRNG / environment work, torch-CPU for the floating-point networks)
of the reference algorithm in /root/reference/mava (liyheng/MAGPO), written by
reading the reference source as text.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline -- the product
path (``magpo_amd``) never imports it and fails loudly without its HIP library.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures and
its dependencies (jax, flax, optax, distrax, tfp, jumanji) are not installed
in the build container, so this restatement is pinned only by
  * the Random123 threefry2x32 known-answer vectors,
  * the hand-worked examples in SURVEY.md Appendix D,
  * algebraic identities of the reference design (recurrent == chunkwise
    retention, GAE vs the O(T^2) direct sum, fp64 finite-difference gradients).
Every function cites the reference file:line it follows.
"""
