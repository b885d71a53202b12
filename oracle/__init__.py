"""CPU restatement of the reference's hot-path operators -- TEST INFRASTRUCTURE ONLY.

Nothing under ``vllm-splitwise_amd/`` imports this package.  It is used by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` as the checker / reported baseline, never as a product code path.

Every function cites the reference file:line whose semantics it restates.  The
restatements are pinned by ``tests/golden/`` (outputs of the reference's own
Python -- its kernel-test references, scheduler, block manager and input
preparation -- captured by ``tools/gen_golden.py``).

Parity status of what has no executable reference counterpart in this build
environment (CUDA kernels cannot run here):
  * fp8 e5m2 conversion: pinned by the published NVIDIA ``__nv_cvt_*`` semantics
    the reference calls (round-to-nearest-even, saturate-to-finite) and
    cross-checked against ``torch.float8_e5m2`` on the non-saturating range.
  * GPTQ / AWQ 4-bit dequant-GEMM: the reference has no test for them
    ("parity unpinned"); the restatement follows the kernels' dequant formulas.
"""
