"""Per-call tuning knobs of the C ABI, held on the HOST side.

libmagpo_hip.so has no setters, reads no environment variables and keeps no state that changes results or buffer
sizes (include/magpo.h, conventions): every knob below is an argument of the entry points it affects.  The host
objects (SableGuider, GruActor) own one ``Tuning`` and pass its fields on every call, so two learners in one process
can run under different settings and a forward / backward pair can never see different values.

Environment variables are read HERE, once, when the default instance is built (A/B measurements from the shell):
    MAGPO_RET_CHUNK=64          retention chunk kernels on 64-token chunks (default 32)
    MAGPO_GRU_SPLIT_BF16=0|1|2  GRU training scans: 2 (default) = the forward scan's recurrent GEMM on bf16 MFMA with operands split in THREE pieces (24
                                mantissa bits = fp32 operands, six products, fp32 accumulate: error against fp64 no larger than the fp32-MFMA scan's,
                                tests/test_kernels_gpu.py::test_gru_scan_bf16_triples_keep_fp32_accuracy), backward on fp32 MFMA; 0 = fp32 MFMA everywhere;
                                1 = pairs (16 mantissa bits, forward and backward; never a default)
    MAGPO_GRU_BLOCK_ROWS=32|64  recurrent rows per workgroup of the fp32 GRU scans (default: by size)
    MAGPO_LINEAR_LDS=0          wave-autonomous dense kernels instead of the shared-tile ones (MAGPO_LINEAR_LDS64=0: KIN = 64 only)
    MAGPO_LINEAR_BF3=1          dense layers with 128 / 192 inputs (four-wave column blocks) on bf16 MFMA with three-piece operand splits (24 mantissa bits,
                                error against fp64 no larger than the fp32-MFMA kernel's, test_linear_bf16_triples_keep_fp32_accuracy).  OPT-IN: on the
                                3x30-50 sweep the runs with it on the actor ended at 90.5 (ten seeds) against 93.4 without
                                (profiles/r03_sweep_return_at_10M.md) -- not understood, so not a default
    MAGPO_WGRAD_FULL=0 / MAGPO_WGRAD_FULL_X=0 / MAGPO_WGRAD_PAD0=0 / MAGPO_WGRAD_G2=1 / MAGPO_WGRAD_GALT=1|2|3 / MAGPO_WGRAD_BF3=1 (128 x 384 on bf16 triples: opt-in,
                                its accumulation error is 1.2 x the fp32-MFMA kernel's)
    MAGPO_ACT_EPW=4|8|16        envs per wave of the fused acting kernel (default: by size)
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class Tuning:
    ret_chunk_tokens: int = 0     # 0 = default (32), 32 or 64: magpo_retention_num_chunks / _chunk_fwd / _chunk_bwd
    gru_split_bf16: int = 2       # magpo_gru_scan_fwd / _bwd (2: forward scan on bf16 triples = fp32 accuracy, see above; 0: fp32 MFMA)
    gru_block_rows: int = 0       # magpo_gru_scan_fwd / _bwd / magpo_gru_carry
    linear_variant: int = 0       # magpo_linear of the guider (bit mask, see include/magpo.h; bit 2 = bf16 triples for KIN 128 / 192)
    actor_linear_variant: int = 0 # magpo_linear of the GRU actor
    wgrad_variant: int = 0        # magpo_wgrad (bit mask)
    act_envs_per_wave: int = 0    # magpo_sable_act dims[11]

    @classmethod
    def from_env(cls, env=None) -> "Tuning":
        e = os.environ if env is None else env
        off = lambda name: e.get(name) not in (None, "") and int(e[name]) == 0
        on = lambda name: e.get(name) not in (None, "") and int(e[name]) != 0
        t = cls()
        t.ret_chunk_tokens = 64 if e.get("MAGPO_RET_CHUNK") == "64" else 0
        t.gru_split_bf16 = int(e["MAGPO_GRU_SPLIT_BF16"]) if e.get("MAGPO_GRU_SPLIT_BF16") in ("0", "1", "2") else cls().gru_split_bf16
        t.gru_block_rows = int(e.get("MAGPO_GRU_BLOCK_ROWS", 0)) if e.get("MAGPO_GRU_BLOCK_ROWS") in ("32", "64") else 0
        base = (1 if off("MAGPO_LINEAR_LDS") else 0) | (2 if off("MAGPO_LINEAR_LDS64") else 0)
        t.linear_variant = base | (4 if on("MAGPO_LINEAR_BF3") else 0)
        t.actor_linear_variant = base | (4 if on("MAGPO_LINEAR_BF3") else 0)
        t.wgrad_variant = ((1 if off("MAGPO_WGRAD_FULL") else 0) | (2 if off("MAGPO_WGRAD_FULL_X") else 0) | (4 if off("MAGPO_WGRAD_PAD0") else 0)
                           | (8 if on("MAGPO_WGRAD_G2") else 0) | ((int(e.get("MAGPO_WGRAD_GALT", 0)) & 3) << 4) | (64 if on("MAGPO_WGRAD_BF3") else 0))
        t.act_envs_per_wave = int(e["MAGPO_ACT_EPW"]) if e.get("MAGPO_ACT_EPW") in ("4", "8", "16") else 0
        return t
