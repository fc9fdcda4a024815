"""Launch programs: a model forward is compiled once per input shape into a flat list of C-ABI
calls over a static workspace (no allocation, no host sync inside a step), which is what makes
the per-step work capturable in a hipGraph and cheap to replay from Python.

Buffers come from a size-keyed pool with explicit release, so the working set of a whole UNet
evaluation stays small enough to live in the 256 MiB Infinity Cache / L2 between producer and
consumer kernels instead of streaming ~1 GB of fresh HBM lines per step.
"""
import ctypes as C
import math
import os

import torch

from . import lib as L
from . import switches


def plan_key(a, m):
    """Shape signature of an igemm problem for the tuned-plan table (m = the row count the plan is made for)."""
    key = f"{m},{a.N},{a.K},{a.a_mode},{a.a_tf},{a.epi},{max(1, a.batch)}"
    if a.b_trans:                 # data-gradient GEMMs of the training step read W^T: tuned separately
        key += ",bt"
    if a.a_mode == L.A_CONV3X3 and (a.stride != 1 or a.upsample):
        key += f",s{a.stride}u{a.upsample}"
    return key


# ---- tuned launch plans ---------------------------------------------------------------------------------------------------------
# ONE file, dsml_thesis_amd/igemm_plans.json, one section per ARITHMETIC / operand form (all produced offline on an MI355X by
# tools/autotune.py / tools/ps_bench.py and merged by tools/merge_plans.py; a static file, so every rank makes the same choice):
#   "f32"        (tile_cfg, splitk) per shape on the f32 matrix cores: LDS-tiled igemm, row GEMM, slab GEMM
#   "bf16x3"     shapes measured FASTER in the fp32-accurate bf16x3 arithmetic (LDMK_COMPUTE_BF16X3) than their best f32 plan
#   "f16x2"      plans measured in the F16X2 arithmetic itself; a shape listed under "bf16x3" only runs in F16X2 too, on that tile
#   "ps_bf16x3"  shapes measured faster on the pre-split tiles (csrc/igemm_ps.hip: operands in the PS layout, LDS-DMA), bf16x3 planes
#   "ps_f16x2"   the same for the two-plane F16X2 form of the pre-split tiles
# Keys are plan_key() strings "M,N,K,a_mode,a_tf,epi,batch[,bt][,s<stride>u<upsample>]"; loaded as {rest of key: [(M, cfg, splitk)]}.
# A/B overrides: LDMK_PLAN_TABLE / LDMK_X3_TABLE / LDMK_H2_TABLE / LDMK_PS_TABLE / LDMK_PS_H2_TABLE name a file that replaces ONE
# section -- either a flat {key: [cfg, splitk]} file (what the tuning tools write) or another merged file, whose section is taken.
_SECTIONS = {"f32": "LDMK_PLAN_TABLE", "bf16x3": "LDMK_X3_TABLE", "f16x2": "LDMK_H2_TABLE", "ps_bf16x3": "LDMK_PS_TABLE",
             "ps_f16x2": "LDMK_PS_H2_TABLE"}
_TABLES = {}
_PLAN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "igemm_plans.json")


def split_enabled():
    """LDMK_COMPUTE_BF16X3 (fp32-accurate GEMMs from six bf16 MFMAs, include/ldmk.h) for the shapes the plan file lists;
    LDMK_SPLIT_BF16=0 keeps every GEMM on the f32 matrix-core form."""
    return switches.get("LDMK_SPLIT_BF16", "1") != "0"


def f16x2_enabled():
    """The F16X2 arithmetic (include/ldmk.h: fp32-accurate products from three fp16 matrix instructions per term, operands scaled
    into fp16's range, a device flag raised when one leaves it) where a kernel offers it; LDMK_F16X2=0 keeps the bf16x3 forms."""
    return split_enabled() and switches.get("LDMK_F16X2", "1") != "0"


def ps_enabled():
    """Pre-split operands (csrc/igemm_ps.hip, tile_cfg 23+): activations written in the PS layout by their producers and moved
    memory -> LDS by LDS-DMA, for the shapes the "ps_*" sections list.  LDMK_PS=0 keeps the round-3 kernels."""
    return split_enabled() and switches.get("LDMK_PS", "1") != "0"


def _section_enabled(section):
    if section == "f32":
        return not switches.get("LDMK_NO_PLAN_TABLE")
    if section == "bf16x3":
        return split_enabled()
    if section == "f16x2":
        return f16x2_enabled()
    if section == "ps_bf16x3":
        return ps_enabled()
    return ps_enabled() and f16x2_enabled()


def reset_tables():
    """Forget the loaded sections (tests and A/B tools that change the environment between programs)."""
    _TABLES.clear()


def table(section):
    """{rest of key: [(M, cfg, splitk)] sorted by M} of one section of the plan file (see above); {} when its arithmetic is off."""
    t = _TABLES.get(section)
    if t is None:
        import json
        t = {}
        path = switches.get(_SECTIONS[section]) or _PLAN_FILE
        if _section_enabled(section) and os.path.exists(path):
            try:
                with open(path) as fh:
                    raw = json.load(fh)
            except Exception:
                raw = {}
            if any(k in raw for k in _SECTIONS):      # a merged file: this section of it
                raw = raw.get(section, {})
            for k, (cfg, sk) in raw.items():
                m, rest = k.split(",", 1)
                t.setdefault(rest, []).append((int(m), int(cfg), int(sk)))
            for v in t.values():
                v.sort()
        _TABLES[section] = t
    return t


# How far (as a ratio of row counts) a tuned plan may be carried when nothing nearer exists (see choose()).  Every plan of a bucket
# is LEGAL for every M (the split depth is bounded by K and the even-tile rule by the epilogue, both part of the key), so this is
# a performance rule only.  Rounds 1-4 stopped at 2: jobs far from the tuned batches (16 and 128) then found no split-arithmetic
# plan and silently ran the f32 program.  Round 5 carries the nearest tuned plan whatever the distance when no section holds a near
# one (A/B: profiles/r05_plan_coverage.txt); LDMK_PLAN_MAX_RATIO=2 restores the old rule.
PLAN_MAX_RATIO = float(switches.get("LDMK_PLAN_MAX_RATIO", "0")) or None
# direct 3x3 convolutions without a table entry run in F16X2 from this K (= 9 C_in) up, on their f32 plan's tile (Program.plan)
H2_CONV_MIN_K = 1440 if switches.get("LDMK_H2_CONV_RULE", "1") != "0" else (1 << 30)


def lookup(section, rest, m, max_ratio=None):
    """(cfg, splitk) of the tuned shape in `section` with key `rest` whose row count is closest to m on a log scale (ties go to the
    smaller M), or None.  max_ratio: give up beyond that factor between the row counts (entries that record a ROUTE decision --
    direct convolution against Winograd -- hold near the measured size only)."""
    rows = table(section).get(rest)
    if not rows or m <= 0:
        return None
    best = min(rows, key=lambda r: (abs(math.log(r[0] / m)), r[0]))
    lim = max_ratio if max_ratio is not None else PLAN_MAX_RATIO
    if lim is not None and max(best[0], m) > lim * min(best[0], m):
        return None
    return best[1], best[2]


NEAR_RATIO = 2.0
# the job batches the split-arithmetic sections were swept at (tools/autotune.py --x3 / --h2, tools/ps_bench.py, tools/conv_ps_bench.py:
# both latent sizes); the f32 section additionally holds batches 1-128
TUNED_JOB_BATCHES = (16, 128)


def far_from_tuned(job_batch):
    """True when a job of this many samples is more than NEAR_RATIO away from every batch the split-arithmetic plans were measured
    at: its shapes then have no near entry in those sections, and Program.plan() carries the nearest one whatever the distance
    (Program.far_plans) instead of falling back to the f32 program.  Near a tuned batch the sections are complete for the job -- a
    shape missing from them LOST its sweep there -- and only near entries count, as in rounds 1-4."""
    return all(max(job_batch, b) > NEAR_RATIO * min(job_batch, b) for b in TUNED_JOB_BATCHES)


def choose(section, rest, m, far=False):
    """The plan `section` offers for shape `rest` at m rows: the entry within NEAR_RATIO of a tuned row count, or -- far: the job
    is far from every tuned batch (far_from_tuned) -- the nearest entry at any distance (64x64x4 at B = 48: 955 -> 1097
    sample-steps/s, 32x32x3 at B = 5 / 7: +5 / +14 %, profiles/r05_plan_coverage.txt)."""
    p = lookup(section, rest, m, NEAR_RATIO)
    if p is None and far:
        p = lookup(section, rest, m)
    return p


def _rest(a, m):
    return plan_key(a, m).split(",", 1)[1]


def plan_table():
    return table("f32")


def x3_table():
    return table("bf16x3")


def h2_table():
    return table("f16x2")


def ps_table():
    return table("ps_bf16x3")


def ps_h2_table():
    return table("ps_f16x2")


def tuned_plan(a, m, far=False):
    """f32 plan of the tuned shape with the same (N, K, prologue, epilogue) and the closest row count, or None (the C++ heuristic
    decides)."""
    return choose("f32", _rest(a, m), m, far)


def x3_plan(a, m, far=False):
    """(cfg, splitk) of the bf16x3 plan for this shape, or None: only shapes measured faster in this arithmetic are listed."""
    return choose("bf16x3", _rest(a, m), m, far)


def h2_plan(a, m, far=False):
    return choose("f16x2", _rest(a, m), m, far)


def ps_plan(rest, m, h2=False, max_ratio=None, far=False):
    """(cfg, splitk) of the pre-split plan for the shape key `rest` ("N,K,mode,tf,epi,batch") at m rows, or None.  h2: the section
    of the F16X2 form."""
    sec = "ps_f16x2" if h2 else "ps_bf16x3"
    return lookup(sec, rest, m, max_ratio) if max_ratio is not None else choose(sec, rest, m, far)


class ArithSites:
    """The F16X2 range flags of a model, one int32 word per SITE.  A site is a group of launches that share split operands -- a
    convolution with its transform / statistics producer, `LN1 -> QKV -> attention -> to_out`, `LN3 -> GEGLU -> ff.net.2` -- named
    by the reference's parameter prefix, so the same layer is the same site in every launch program of the model (any batch, any
    shape).  Every F16X2 launch of the site points its `range_flag` at the site's word (the ABI has always taken the pointer per
    launch); an out-of-range operand raises it and is saturated (csrc/ldmk_common.h: h2_clamp), so the rest of the evaluation stays
    finite and raises only its own flags.  `raised()` is ONE device -> host copy; the model then DENIES the named sites -- they
    run in the bf16x3 arithmetic from the next program build on -- and leaves the others in F16X2."""

    MAX_SITES = 1024

    def __init__(self, device):
        self.flags = torch.zeros(self.MAX_SITES, device=device, dtype=torch.int32)
        self.index = {}
        self.denied = set()

    def flag(self, name):
        """The site's flag word (a 1-element view), or None when the site has been denied the F16X2 arithmetic."""
        if name in self.denied:
            return None
        i = self.index.get(name)
        if i is None:
            i = self.index[name] = len(self.index)
            assert i < self.MAX_SITES, "more arithmetic sites than flag words"
        return self.flags[i:i + 1]

    def raised(self, clear=True):
        """Names of the sites whose flag is up (one host sync); the words are zeroed again when any was."""
        n = len(self.index)
        if n == 0:
            return []
        up = torch.nonzero(self.flags[:n].cpu()).flatten().tolist()
        if not up:
            return []
        if clear:
            self.flags.zero_()
        names = [None] * n
        for k, i in self.index.items():
            names[i] = k
        return [names[i] for i in up]


class Program:
    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = L.load()
        L.init(self.device.index if self.device.index is not None else torch.cuda.current_device())
        self.calls = []          # (fn, args, keepalive)
        self._free = {}          # numel -> [tensor]
        self._all = []
        self.inputs = {}
        self.outputs = {}

    # ---- workspace -----------------------------------------------------------------------
    def alloc(self, *shape, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= int(s)
        key = (n, dtype)
        lst = self._free.get(key)
        if lst:
            return lst.pop().view(*shape)
        t = torch.empty(n, device=self.device, dtype=dtype)
        self._all.append(t)
        return t.view(*shape)

    def release(self, *tensors):
        for t in tensors:
            if t is None:
                continue
            self._free.setdefault((t.numel(), t.dtype), []).append(t.reshape(-1))

    def alloc_ps(self, rows, k, batch=1):
        """A pool buffer for a [rows][k] matrix (x batch) in the PS layout (include/ldmk.h): uint8 [batch * ldmk_ps_bytes]; the
        two-plane F16X2 form while the program runs in that arithmetic (h2_flag)."""
        nb = (self.lib.ldmk_ps_bytes_h2 if getattr(self, "h2_flag", None) is not None else self.lib.ldmk_ps_bytes)(int(rows), int(k))
        assert nb > 0, (rows, k)
        return self.alloc(int(batch) * nb, dtype=torch.uint8)

    def workspace_bytes(self):
        return sum(t.numel() * t.element_size() for t in self._all)

    # ---- calls ---------------------------------------------------------------------------
    def add(self, name, *args, keep=None):
        self.calls.append((getattr(self.lib, name), args, keep, name))

    def splitk_workspace(self, elems):
        """One scratch for split-K partial slabs, shared by every GEMM of the program (stream-ordered)."""
        if getattr(self, "_skws", None) is None or self._skws.numel() < elems:
            self._skws = torch.empty(int(elems), device=self.device, dtype=torch.float32)
            self._all.append(self._skws)
            for _, _, a, name in self.calls:          # re-point GEMMs recorded against an older scratch
                if name == "ldmk_igemm" and a.splitk_ws and not hasattr(a, "_own_slabs"):
                    a.splitk_ws, a.splitk_ws_elems = self._skws.data_ptr(), self._skws.numel()
        return self._skws

    def splitk_counters(self, n=16384):
        """Arrival counters of the in-launch split-K combine: zeroed once; every GEMM launch leaves them zeroed, and the
        GEMMs of a program are stream-ordered, so they all share this one array."""
        if getattr(self, "_skcnt", None) is None:
            self._skcnt = torch.zeros(n, device=self.device, dtype=torch.int32)
            self._all.append(self._skcnt)
        return self._skcnt

    def plan(self, args, scale_m=None, allow_splitk=True, batch_is_samples=True, per_sample=False):
        """Choose (tile shape, K split) for a GEMM and pin them in `args`: for the real M, or -- with scale_m = (num, den) --
        for M*num/den rows, which pins the K-summation order so that results are bitwise independent of how a batch is
        split across calls / ranks.  Returns (tile_cfg, splitk).
        per_sample: the GEMM is one problem PER SAMPLE (args.batch = samples) also when this call happens to hold one sample: a
        one-sample shard of a job must plan it as the job does -- a batch of problems, never looked up in the tables -- not as a
        plain GEMM with more rows."""
        m, nbatch = args.M, args.batch
        sample_batch = per_sample or (nbatch > 1 and batch_is_samples)
        if scale_m is not None and scale_m[0] != scale_m[1]:
            if sample_batch:                       # batched GEMM (one problem per sample): the job-wide view has more problems
                args.batch = max(1, max(1, nbatch) * scale_m[0] // scale_m[1])
            else:
                args.M = max(1, m * scale_m[0] // scale_m[1])
        cfg, sk = C.c_int(0), C.c_int(0)
        # bf16x3 arithmetic for the shapes measured faster in it (needs the weight's split images; decided on the policy row
        # count like every plan, so a sample's result does not depend on how the batch is sharded)
        if (args.compute == L.COMPUTE_F32 and not args.b_trans and not args.raw_slabs and not sample_batch
                and args.M > 0 and (x3_table() or h2_table())):
            h2_flag = getattr(self, "h2_flag", None)
            far = bool(getattr(self, "far_plans", False))      # the job is far from every tuned batch: carry the nearest plan (choose)
            hp = h2_plan(args, args.M, far) if h2_flag is not None else None       # a plan measured in the F16X2 arithmetic itself
            xp = hp if hp is not None else x3_plan(args, args.M, far)
            if xp is None and h2_flag is not None and args.a_mode == L.A_CONV3X3 and args.K >= H2_CONV_MIN_K:
                # A 3x3 convolution nobody measured in a split arithmetic (the tables hold the shapes of the B = 16 / 128 jobs; at
                # other batches the low-resolution convolutions leave the Winograd route and show up as direct ones with
                # K = 9 C_in of 2880-11520): long-K implicit GEMMs are matrix-bound, where three fp16 MFMAs per product beat eight
                # f32 ones on every tile -- they take the F16X2 arithmetic on the tile and K split of their f32 plan
                # (A/B at 64x64x4 B = 2 and 32x32x3 B = 5 / 7: profiles/r05_plan_coverage.txt; LDMK_H2_CONV_RULE=0 turns it off)
                fp = tuned_plan(args, args.M, far)
                if fp is None:
                    c_, k_ = C.c_int(0), C.c_int(0)
                    keep = (args.splitk_ws, args.splitk_ws_elems)
                    if allow_splitk:
                        args.splitk_ws, args.splitk_ws_elems = 1, 1 << 40
                    self.lib.ldmk_igemm_plan(C.byref(args), C.byref(c_), C.byref(k_))
                    args.splitk_ws, args.splitk_ws_elems = keep
                    fp = (c_.value, max(1, k_.value))
                if int(fp[0]) in (1, 2, 4, 5):
                    xp = (int(fp[0]), int(fp[1]))
            if xp is not None:
                from . import ops as _ops
                saved = (args.M, args.batch, args.tile_cfg, args.splitk, args.splitk_ws, args.splitk_ws_elems)
                args.M, args.batch = m, nbatch
                if int(xp[0]) > 6:
                    args.a_split, args.a_split_ld = 0, 0          # the warp-specialised tiles split A themselves
                # the F16X2 arithmetic (three fp16 products per term, include/ldmk.h) while the owner's range flag is down:
                # same shapes, the LDS-tiled form of the tile (the warp-specialised 256-row tiles map to 128x160 / 128x128)
                ok = h2_flag is not None and not args.a_split and _ops.set_split_h2(args, h2_flag)
                if ok:
                    xp = ({21: 5, 22: 1}.get(int(xp[0]), int(xp[0])), xp[1])
                else:
                    ok = _ops.set_split(args)
                if ok:
                    # legal for the POLICY problem first (the row count every shard of the job shares: a tile that only the
                    # smaller real problem could run must not be chosen, or a shard and the whole job would differ), then for
                    # the real one
                    args.tile_cfg, args.splitk = int(xp[0]), int(xp[1])
                    args.splitk_ws, args.splitk_ws_elems = 1, 1 << 40
                    args.M, args.batch = saved[0], saved[1]
                    ok = self.lib.ldmk_igemm_check(C.byref(args)) == 0
                    args.M, args.batch = m, nbatch
                    ok = ok and self.lib.ldmk_igemm_check(C.byref(args)) == 0
                (args.M, args.batch, args.tile_cfg, args.splitk, args.splitk_ws, args.splitk_ws_elems) = saved
                if ok:
                    args.M, args.batch = m, nbatch
                    args.tile_cfg, args.splitk = int(xp[0]), max(1, int(xp[1]))
                    args.splitk_ws, args.splitk_ws_elems = 0, 0
                    if args.tile_cfg > 6:
                        args.a_split, args.a_split_ld = 0, 0      # the warp-specialised tiles split A themselves
                    return args.tile_cfg, args.splitk
                args.compute, args.w_split, args.w_split_ld, args.w_split_bstride = L.COMPUTE_F32, 0, 0, 0
                args.w_scale_exp, args.range_flag = 0, 0
        # (a batch that is not per sample -- the 16 transform positions of a Winograd convolution -- is part of the plan key)
        tuned = tuned_plan(args, args.M, bool(getattr(self, "far_plans", False))) if not sample_batch and args.M > 0 else None
        if tuned is not None and tuned[0] > 6:
            # a row-GEMM wave tile (7..12, never splits K) or a slab-GEMM shape (13..20, small row counts): legal only with
            # the fragment-order weight copy and when the tile divides this problem (per-sample operands need
            # rows_per_sample % tile rows == 0); else the heuristic decides
            saved = (args.M, args.batch, args.tile_cfg, args.splitk, args.splitk_ws, args.splitk_ws_elems, args.raw_slabs)
            args.tile_cfg = int(tuned[0])
            args.splitk = 1 if tuned[0] <= 12 else int(tuned[1])
            args.splitk_ws, args.splitk_ws_elems = 1, 1 << 40          # (validated against the real scratch below)
            if args.splitk == 1:
                args.raw_slabs = 0                                      # an unsplit plan writes its output itself
            ok = bool(args.w_frag) and self.lib.ldmk_igemm_check(C.byref(args)) == 0      # ... for the policy problem (see above)
            args.M, args.batch = m, nbatch
            if not ok or self.lib.ldmk_igemm_check(C.byref(args)) != 0:                  # ... and for the real one
                tuned = None
            (args.M, args.batch, args.tile_cfg, args.splitk, args.splitk_ws, args.splitk_ws_elems, args.raw_slabs) = saved
        if tuned is not None:
            cfg.value, sk.value = int(tuned[0]), int(tuned[1])
        else:
            if allow_splitk:
                # plan against a generous virtual scratch, then size the real one to what was chosen
                args.splitk_ws, args.splitk_ws_elems = 1, 1 << 40
            self.lib.ldmk_igemm_plan(C.byref(args), C.byref(cfg), C.byref(sk))
        args.M, args.batch = m, nbatch
        args.tile_cfg, args.splitk = cfg.value, max(1, sk.value)
        args.splitk_ws, args.splitk_ws_elems = 0, 0
        args.a_split, args.a_split_ld = 0, 0          # (pre-split A belongs to the bf16x3 LDS-tiled plans, returned above)
        return args.tile_cfg, args.splitk

    def igemm(self, args, scale_m=None, allow_splitk=True, batch_is_samples=True, per_sample=False):
        """Record a GEMM (planned here, once: see plan()); a split-K plan gets the program's shared scratch and is
        followed by its reduce launch."""
        self.plan(args, scale_m, allow_splitk, batch_is_samples, per_sample)
        if args.splitk > 1:
            need = max(1, args.batch) * args.splitk * args.M * args.N
            ws = self.splitk_workspace(need)
            args.splitk_ws, args.splitk_ws_elems = ws.data_ptr(), ws.numel()
            # In-launch combine (each tile's last-arriving workgroup sums the slabs) is built and tested, but OFF: split-K is
            # chosen exactly when a GEMM has few output tiles, so the combine runs on those few workgroups (10 of 256 CUs
            # for the 8x8-level convolutions at batch 1) reading `splitk` slabs with sc1 traffic, while the reduce launch
            # spreads the same bytes over the whole chip.  Measured A/B (sample-steps/s, reduce launch -> in-launch):
            # first version (one atomic load per element) 476 -> 446 at 64x64x4 B=16, 1657 -> 1231 at 32x32x3, 249 -> 141
            # at B=1; with the loads batched (64 in flight per wave) 1720 -> 1614 at 32x32x3 and 252 -> 249 at B=1.
            # LDMK_SPLITK_IN_LAUNCH=1 turns it on.
            if switches.get("LDMK_SPLITK_IN_LAUNCH"):
                cnt = self.splitk_counters()
                args.splitk_counters, args.splitk_counters_len = cnt.data_ptr(), cnt.numel()
        self.calls.append((self.lib.ldmk_igemm, (C.byref(args),), args, "ldmk_igemm"))

    def igemm_ps(self, args, cfg, sk):
        """Record a GEMM on a pre-split tile (args carry a_ps / w_ps; the plan comes from the PS table, decided by the caller
        BEFORE it had the producer write the A operand in that layout)."""
        args.tile_cfg, args.splitk = int(cfg), max(1, int(sk))
        assert args.compute in (L.COMPUTE_BF16X3, L.COMPUTE_F16X2)       # (set by ops.make_igemm_args from the form of w_ps)
        if args.splitk > 1:
            ws = self.splitk_workspace(max(1, args.batch) * args.splitk * args.M * args.N)
            args.splitk_ws, args.splitk_ws_elems = ws.data_ptr(), ws.numel()
        rc = self.lib.ldmk_igemm_check(C.byref(args))
        if rc != 0:
            L.check(rc, "ldmk_igemm_check (pre-split plan)")
        self.calls.append((self.lib.ldmk_igemm, (C.byref(args),), args, "ldmk_igemm"))

    def igemm_raw(self, args, slabs):
        """Record an already planned GEMM that leaves its raw K slabs [splitk][M][N] in `slabs` for the consumer
        (ldmk_post / ldmk_attn_self_small) -- no reduce launch; with splitk == 1 the one 'slab' is simply the GEMM's output
        (the caller passes no bias / residual, so that output is the raw product)."""
        if args.splitk > 1:
            args.raw_slabs = 1
            args.splitk_ws, args.splitk_ws_elems = slabs.data_ptr(), args.splitk * args.M * args.N
        else:
            args.raw_slabs = 0
            args.out = slabs.data_ptr()
        args._own_slabs = slabs                       # (keeps the buffer alive; marks the call for splitk_workspace below)
        self.calls.append((self.lib.ldmk_igemm, (C.byref(args),), args, "ldmk_igemm"))

    def post(self, pargs, keep=None):
        need = self.lib.ldmk_post_scratch_elems(C.byref(pargs))
        if need > 0:                                   # row-tiled GroupNorm (large images): one shared, stream-ordered scratch
            if getattr(self, "_gnws", None) is None or self._gnws.numel() < need:
                self._gnws = torch.empty(int(need), device=self.device, dtype=torch.float32)
                self._all.append(self._gnws)
                for _, _, k, name in self.calls:
                    if name == "ldmk_post" and k[0].gn_scratch:
                        k[0].gn_scratch, k[0].gn_scratch_elems = self._gnws.data_ptr(), self._gnws.numel()
            pargs.gn_scratch, pargs.gn_scratch_elems = self._gnws.data_ptr(), self._gnws.numel()
        self.calls.append((self.lib.ldmk_post, (C.byref(pargs),), (pargs, keep), "ldmk_post"))

    def run(self, stream=None):
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        for fn, args, _, name in self.calls:
            rc = fn(*args, st)
            if rc != 0:
                L.check(rc, name)


class NetBuilder:
    """Emits the recurring layer patterns of the UNet / VQGAN into a Program (NHWC activations)."""

    def __init__(self, pg, n, pin=None, sites=None):
        from . import ops
        self.pg, self.n, self.pin, self.ops = pg, n, pin, ops
        self._stats = {}          # tensor data_ptr -> GroupNorm partial records [rows/32][C][3]
        # F16X2 is decided per SITE (ArithSites): inside `with nb.site(name):` the builder's and the program's `h2_flag` are the
        # site's flag word -- or None for a denied site, and outside any site -- and every emitter below reads them there
        self.sites = sites
        self.h2_flag = getattr(pg, "h2_flag", None)
        pg.h2_flag = self.h2_flag

    def site(self, name):
        """Context in which launches are emitted in the arithmetic of site `name`: F16X2 with the site's range flag, or -- the
        model runs without F16X2, or this site was denied it after its flag went up -- the bf16x3 / f32 forms."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            if self.sites is None:          # (a builder driven by hand -- tests, tools -- keeps whatever flag its caller set)
                yield self.h2_flag
                return
            old = (self.h2_flag, self.pg.h2_flag)
            self.h2_flag = self.pg.h2_flag = self.sites.flag(name)
            try:
                yield self.h2_flag
            finally:
                self.h2_flag, self.pg.h2_flag = old
        return ctx()

    @staticmethod
    def ptr(t):
        return 0 if t is None else (t if isinstance(t, int) else t.data_ptr())

    # ---- GroupNorm statistics: partial records are produced by the tensor's producer (igemm epilogue) when
    # possible and cached per tensor, so a tensor normalised twice (ResBlock output -> SpatialTransformer norm ->
    # later a skip-concat GroupNorm) is never re-read for statistics.
    def stats_buffer(self, rows, c):
        return self.pg.alloc(rows // 32, c, 3)

    def release(self, *tensors):
        """Return buffers to the pool; statistics cached for them die with them (the pool recycles addresses)."""
        for t in tensors:
            if t is not None:
                self.drop_stats(t)
        self.pg.release(*tensors)

    def attach_stats(self, t, partial):
        self._stats[t.data_ptr()] = partial

    def drop_stats(self, t):
        p = self._stats.pop(t.data_ptr(), None)
        if p is not None:
            self.pg.release(p)

    def _partial(self, x, hw):
        p = self._stats.get(x.data_ptr())
        if p is None:
            c = x.shape[-1]
            p = self.pg.alloc(self.n * self.pg.lib.ldmk_gn_chunks(hw), c, 3)
            self.pg.add("ldmk_gn_partial", self.ptr(x), c, self.n, hw, self.ptr(p))
            self._stats[x.data_ptr()] = p
        return p

    def gn(self, x0, x1, hw, gamma, beta, eps, film=None):
        """GroupNorm(32) statistics of (the channel concat of) NHWC tensors -> coef planes [n][2][C].  film = (pointer to a
        per-sample [n][ld] vector (scale | shift), ld): use_scale_shift_norm's `norm(h) (1 + scale) + shift` folded into the planes."""
        pg, p_ = self.pg, self.ptr
        c0 = x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        pa = self._partial(x0, hw)
        pb = None if x1 is None else self._partial(x1, hw)
        coef = pg.alloc(self.n, 2, c0 + c1)
        pg.add("ldmk_gn_finalize", p_(pa), c0, p_(pb), c1, self.n, hw, 32, eps, p_(gamma), p_(beta), p_(coef))
        if film is not None:
            pg.add("ldmk_gn_coef_film", p_(coef), film[0], film[1], self.n, c0 + c1)
        return coef

    def gn_act(self, x0, x1, hw, gamma, beta, eps, silu=True, film=None):
        """GroupNorm(32) [+SiLU] of (the concat of) NHWC tensors materialised once: statistics pass + one
        elementwise pass.  Returns a contiguous [n*hw][C] tensor for the consumer GEMM to read raw."""
        pg, p_ = self.pg, self.ptr
        c0 = x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        coef = self.gn(x0, x1, hw, gamma, beta, eps, film)
        y = pg.alloc(self.n * hw, c0 + c1)
        pg.add("ldmk_gn_apply", p_(x0), c0, p_(x1), c1, p_(coef), p_(y), self.n, hw, 1 if silu else 0)
        self.release(coef)
        return y

    def conv(self, x0, x1, wp, bias, h, w, coef=None, stride=1, pad_lo=1, upsample=False, batch_vec=None, bv_ld=0,
             residual=None, out=None, stats=False, wf=None):
        """3x3 conv (implicit GEMM) with optional GN+SiLU prologue / per-sample vector / residual epilogue."""
        pg, n, ops = self.pg, self.n, self.ops
        c0 = x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        cout = wp.shape[1]
        if upsample:
            oh, ow = 2 * h, 2 * w
        elif pad_lo == 1:
            oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        else:                                   # asymmetric (0,1,0,1) padding, model.py:72-76
            oh, ow = (h + 1 - 3) // stride + 1, (w + 1 - 3) // stride + 1
        if out is None:
            out = pg.alloc(n, oh, ow, cout)
        a = ops.make_igemm_args(n * oh * ow, cout, 9 * (c0 + c1), x0, c0, wp, out, cout, oh * ow, a1=x1, c1=c1,
                                conv=(h, w, oh, ow, stride, pad_lo, 1 if upsample else 0),
                                tf=L.TF_NONE if coef is None else L.TF_AFFINE_SILU, tf_coef=coef, bias=bias,
                                residual=residual, w_frag=wf)
        if batch_vec is not None:
            a.batch_vec, a.batch_vec_ld = self.ptr(batch_vec), bv_ld
        self._maybe_stats(a, out.view(-1, cout), oh * ow, stats)
        pg.igemm(a, self.pin)
        return out

    # ---- GroupNorm -> SiLU -> Conv2d 3x3 (stride 1), the ResBlock pattern.  Large enough problems go through Winograd
    # F(2x2,3x3) (csrc/winograd.hip): 4 instead of 9 multiplications per output and input channel on a path whose matrix
    # work is power-limited; the transforms replace the gn_apply pass and the convolution's epilogue.  The decision is made on
    # the plan-policy batch (pin[0]), like the tile plans, so a sample's result does not depend on how a batch is sharded.
    # Measured on MI355X (tools/winograd_bench.py, in-step times of the direct form): wins from 320 input channels and
    # 256 tiles up (640->640 at 16x16, B = 16: 243+8 -> 165 us); loses at 160 channels (transform traffic) and at batch 1.
    # (the two environment overrides exist for A/B experiments only).  Tiles: 1024 -> 256 is +2.3 % at 32x32x3 B = 16 (the
    # 640-channel level has 256 tiles there); 64 tiles (batch 1) loses 7 %.
    WINO_MIN_TILES = int(switches.get("LDMK_WINO_MIN_TILES", "256"))
    UP_MIN_PIXELS = 1024
    WINO_MIN_CIN = int(switches.get("LDMK_WINO_MIN_CIN", "320"))

    def winograd_ok(self, cin, h, w):
        import os
        if switches.get("LDMK_NO_WINOGRAD"):
            return False
        pol_n = self.pin[0] if self.pin else self.n
        return (cin >= self.WINO_MIN_CIN and pol_n * (h // 2) * (w // 2) >= self.WINO_MIN_TILES and h % 2 == 0
                and w in (8, 16, 32, 64, 128) and (h * w) % 32 == 0 and (w >= 16 or (h // 2) % 2 == 0))

    def ps_query_conv(self, M, N, K, stride=1):
        """(tile_cfg, splitk) of the CONV-MODE pre-split tile (csrc/igemm_ps.hip: igemm_psc_kernel, F16X2 only) for a 3x3
        convolution of this shape, at the plan-policy row count -- or None.  Listed in the "ps_f16x2" section under the conv key
        (a_mode 1) when the direct convolution on pre-split operands was measured faster in the step than the shape's other route
        (the in-register-split implicit GEMM for the 160-channel convolutions, Winograd GEMM + both transforms for the wide ones:
        profiles/r05_ab_conv_ps.txt).  LDMK_PSC=0 turns the route off; LDMK_PSC_FORCE="cfg,splitk" forces it for every eligible
        convolution (A/B runs)."""
        if not ps_enabled() or self.h2_flag is None or switches.get("LDMK_PSC", "1") == "0":
            return None
        forced = switches.get("LDMK_PSC_FORCE")
        if forced:
            cfg, sk = (int(v) for v in forced.split(","))
            cin32 = K // (9 * 32)
            while cin32 % sk:
                sk -= 1
            return cfg, sk
        m = M
        if self.pin is not None and self.pin[0] != self.pin[1]:
            m = max(1, M * self.pin[0] // self.pin[1])
        return ps_plan(f"{N},{K},{L.A_CONV3X3},0,0,1" + ("" if stride == 1 else f",s{stride}u0"), m, h2=True, max_ratio=2.0)

    def gn_conv(self, x0, x1, h, w, gamma, beta, eps, wp, u, bias, batch_vec=None, bv_ld=0, residual=None, out=None,
                stats=False, wf=None, u_ps=None, wp_ps=None, film=None):
        """GroupNorm(32)+SiLU of (the concat of) x0 | x1, then the 3x3 convolution with packed weights `wp` (implicit GEMM)
        or, when `u` (ops.pack_winograd) is given and the problem is large enough, through Winograd -- or, `wp_ps`
        (ops.pack_wps(wp, h2=True)) given and the shape listed (ps_query_conv), as a direct convolution on the conv-mode
        pre-split tile: the GroupNorm-apply pass writes the activation once in the PS layout and the nine taps are LDS-DMA
        address sets into it."""
        pg, n, ops, p_ = self.pg, self.n, self.ops, self.ptr
        c0 = x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        cin = c0 + c1
        cout_ = wp.shape[1]
        plan_c = (self.ps_query_conv(n * h * w, cout_, 9 * cin)
                  if (wp_ps is not None and cin % 32 == 0 and cout_ % 32 == 0 and c0 % 8 == 0 and c1 % 8 == 0) else None)
        if plan_c is not None:
            hf = self.h2_flag
            coef = self.gn(x0, x1, h * w, gamma, beta, eps, film)
            y_ps = pg.alloc_ps(n * h * w, cin)
            pg.add("ldmk_gn_apply_ps_h2", p_(x0), c0, p_(x1), c1, p_(coef), p_(y_ps), n, h * w, 1, p_(hf))
            self.release(coef)
            if out is None:
                out = pg.alloc(n, h, w, cout_)
            a = ops.make_igemm_args(n * h * w, cout_, 9 * cin, None, cin, wp, out, cout_, h * w, conv=(h, w, h, w, 1, 1, 0), bias=bias,
                                    residual=residual, a_ps=y_ps, w_ps=wp_ps, range_flag=hf)
            if batch_vec is not None:
                a.batch_vec, a.batch_vec_ld = self.ptr(batch_vec), bv_ld
            a._algo_flops = 2.0 * (n * h * w) * cout_ * (9 * cin)
            self._maybe_stats(a, out.view(-1, cout_), h * w, stats)
            pg.igemm_ps(a, *plan_c)
            pg.release(y_ps)
            return out
        if u is None or not self.winograd_ok(cin, h, w):
            y = self.gn_act(x0, x1, h * w, gamma, beta, eps, film=film)
            res = self.conv(y.view(n, h, w, cin), None, wp, bias, h, w, batch_vec=batch_vec, bv_ld=bv_ld, residual=residual,
                            out=out, stats=stats, wf=wf)
            self.release(y)
            return res
        cout = u.shape[2]
        tiles = n * (h // 2) * (w // 2)
        coef = self.gn(x0, x1, h * w, gamma, beta, eps, film)
        Mb = pg.alloc(16, tiles, cout)
        if out is None:
            out = pg.alloc(n, h, w, cout)
        # with a pre-split plan for the 16 plane GEMMs (csrc/igemm_ps.hip) the input transform writes V in the PS layout: same
        # matrix, split once here instead of once per N-tile inside the GEMM
        plan = self.ps_query(tiles, cout, cin, batch=16) if (u_ps is not None and c0 % 16 == 0 and c1 % 16 == 0) else None
        if plan is not None:
            V = pg.alloc_ps(tiles, cin, batch=16)
            hf = getattr(pg, "h2_flag", None)          # (a program in the F16X2 arithmetic: V as two fp16 planes, range-checked)
            if hf is not None:
                pg.add("ldmk_winograd_input_ps_h2", p_(x0), c0, p_(x1), c1, p_(coef), 1, n, h, w, p_(V), p_(hf))
            else:
                pg.add("ldmk_winograd_input_ps", p_(x0), c0, p_(x1), c1, p_(coef), 1, n, h, w, p_(V))
            a = ops.make_igemm_args(tiles, cout, cin, None, cin, u, Mb, cout, tiles, batch=16, w_bstride=cin * cout,
                                    out_bstride=tiles * cout, a_ps=V.view(16, -1), w_ps=u_ps, range_flag=hf)
        else:
            V = pg.alloc(16, tiles, cin)
            pg.add("ldmk_winograd_input", p_(x0), c0, p_(x1), c1, p_(coef), 1, n, h, w, p_(V))
            a = ops.make_igemm_args(tiles, cout, cin, V, cin, u, Mb, cout, tiles, batch=16, a_bstride=tiles * cin,
                                    w_bstride=cin * cout, out_bstride=tiles * cout)
        a._winograd = True            # (tools/autotune.py sweeps these batched problems; per-sample batches are not planned by table)
        a._algo_flops = 2.0 * (4 * tiles) * cout * (9 * cin)      # the direct convolution's arithmetic (bench.py)
        if plan is not None:
            pg.igemm_ps(a, *plan)
        else:
            pg.igemm(a, self.pin, batch_is_samples=False)
        part = None
        out2d = out.view(-1, cout)
        if stats:
            self.drop_stats(out2d)
            part = self.stats_buffer(n * h * w, cout)
        pg.add("ldmk_winograd_output", p_(Mb), p_(bias), p_(batch_vec), bv_ld, p_(residual), p_(out), p_(part), n, h, w, cout)
        if stats:
            self.attach_stats(out2d, part)
        self.release(coef)
        pg.release(V, Mb)
        return out

    def up_conv(self, x, h, w, wp, w4, bias, stats=False, w4_ps=None):
        """Upsample (nearest x2) + Conv2d 3x3: the implicit GEMM with the upsampling folded into its gather, or -- `w4`
        (ops.pack_upconv) given, >= 320 channels, >= UP_MIN_PIXELS low-resolution pixels at the plan-policy batch -- four 2x2-tap phase
        convolutions on the low-resolution input: 4/9 of the multiplications, exact (measured: 898 -> 522 us for 640->640 at
        16x16 -> 32x32, B = 16)."""
        import os
        pg, n, ops, p_ = self.pg, self.n, self.ops, self.ptr
        c = x.shape[-1]
        pol_n = self.pin[0] if self.pin else n
        if (w4 is None or switches.get("LDMK_NO_WINOGRAD") or c < self.WINO_MIN_CIN or pol_n * h * w < self.UP_MIN_PIXELS
                or w not in (8, 16, 32, 64)):
            return self.conv(x, None, wp, bias, h, w, upsample=True, stats=stats)
        cout = w4.shape[2]
        pix = n * h * w
        Pm = pg.alloc(4, pix, cout)
        out = pg.alloc(n, 2 * h, 2 * w, cout)
        plan = self.ps_query(pix, cout, 4 * c, batch=4) if (w4_ps is not None and c % 16 == 0) else None
        if plan is not None:          # the gather writes its phase operands in the PS layout (csrc/igemm_ps.hip)
            A = pg.alloc_ps(pix, 4 * c, batch=4)
            hf = getattr(pg, "h2_flag", None)
            if hf is not None:
                pg.add("ldmk_upconv_gather_ps_h2", p_(x), c, n, h, w, p_(A), p_(hf))
            else:
                pg.add("ldmk_upconv_gather_ps", p_(x), c, n, h, w, p_(A))
            a = ops.make_igemm_args(pix, cout, 4 * c, None, 4 * c, w4, Pm, cout, pix, batch=4, w_bstride=4 * c * cout,
                                    out_bstride=pix * cout, a_ps=A.view(4, -1), w_ps=w4_ps, range_flag=hf)
        else:
            A = pg.alloc(4, pix, 4 * c)
            pg.add("ldmk_upconv_gather", p_(x), c, n, h, w, p_(A))
            a = ops.make_igemm_args(pix, cout, 4 * c, A, 4 * c, w4, Pm, cout, pix, batch=4, a_bstride=pix * 4 * c,
                                    w_bstride=4 * c * cout, out_bstride=pix * cout)
        a._winograd = True            # a batch of transform planes, not of samples: planned by table like the Winograd GEMMs
        a._algo_flops = 2.0 * (4 * pix) * cout * (9 * c)
        if plan is not None:
            pg.igemm_ps(a, *plan)
        else:
            pg.igemm(a, self.pin, batch_is_samples=False)
        part = None
        out2d = out.view(-1, cout)
        if stats:
            self.drop_stats(out2d)
            part = self.stats_buffer(4 * pix, cout)
        pg.add("ldmk_upconv_scatter", p_(Pm), p_(bias), p_(out), p_(part), n, h, w, cout)
        if stats:
            self.attach_stats(out2d, part)
        pg.release(A, Pm)
        return out

    def _maybe_stats(self, a, out2d, rows_per_sample, stats):
        if stats and a.M % 32 == 0 and rows_per_sample % 32 == 0:
            self.drop_stats(out2d)
            part = self.stats_buffer(a.M, a.N)
            a.stats_out = part.data_ptr()
            self.attach_stats(out2d, part)

    def ps_query(self, M, N, K, tf=L.TF_NONE, epi=L.EPI_NONE, batch=1, per_sample=True):
        """(tile_cfg, splitk) of the pre-split plan (csrc/igemm_ps.hip) for a rows-mode GEMM of this shape, looked up at the
        plan-policy row count like every plan -- or None.  Asked BEFORE the producer of the A operand is emitted: with a plan the
        producer writes the operand in the PS layout."""
        if not ps_enabled():
            return None
        m = M
        if self.pin is not None and self.pin[0] != self.pin[1] and per_sample:
            m = max(1, M * self.pin[0] // self.pin[1])
        # (a program in the F16X2 arithmetic has its own table: its operands are two fp16 planes, not three bf16 ones)
        return ps_plan(f"{N},{K},{L.A_ROWS},{tf},{epi},{max(1, batch)}", m, h2=getattr(self.pg, "h2_flag", None) is not None,
                       far=bool(getattr(self.pg, "far_plans", False)))

    def lin_ps(self, plan, M, K, a_ps, wp, w_ps, bias, rows_per_sample, out=None, out_ps=None, geglu=False, stats=False, **kw):
        """Linear on a pre-split tile: a_ps = the [M][K] input in the PS layout, w_ps = ops.pack_wps(wp).  `out` None with out_ps
        given: the result exists in the PS layout only (its one consumer is another pre-split GEMM)."""
        pg, ops = self.pg, self.ops
        N = wp.shape[1]
        ncol = N // 2 if geglu else N
        if out is None and out_ps is None:
            out = pg.alloc(M, ncol)
        a = ops.make_igemm_args(M, N, K, None, K, wp, out, ncol, rows_per_sample, bias=bias,
                                epi=L.EPI_GEGLU if geglu else L.EPI_NONE, a_ps=a_ps, w_ps=w_ps, out_ps=out_ps,
                                range_flag=getattr(pg, "h2_flag", None), **kw)
        if out is not None:
            self._maybe_stats(a, out, rows_per_sample, stats)
        pg.igemm_ps(a, *plan)
        return out

    def lin(self, x0, wp, bias, rows_per_sample, x1=None, out=None, geglu=False, stats=False, wf=None, **kw):
        """Linear / 1x1 conv on token rows, with the igemm prologue/epilogue options passed through.  `wf`: the
        fragment-order copy of wp (ops.pack_wfrag); with it the plan may pick the wave-autonomous row GEMM."""
        pg, ops = self.pg, self.ops
        M, c0 = x0.shape[0], x0.shape[-1]
        c1 = 0 if x1 is None else x1.shape[-1]
        N = wp.shape[1]
        ncol = N // 2 if geglu else N
        if out is None:
            out = pg.alloc(M, ncol)
        a = ops.make_igemm_args(M, N, c0 + c1, x0, c0, wp, out, ncol, rows_per_sample, a1=x1, c1=c1, bias=bias,
                                epi=L.EPI_GEGLU if geglu else L.EPI_NONE, w_frag=wf, **kw)
        self._maybe_stats(a, out, rows_per_sample, stats)
        pg.igemm(a, self.pin)
        return out


class GraphedProgram:
    """A Program (or any callable that only enqueues on the current stream) captured into a hipGraph
    through torch.cuda.CUDAGraph (hipStreamBeginCapture/hipGraphLaunch underneath)."""

    def __init__(self, fn, warmup=2):
        self.fn = fn
        # the warm-up launches may draw random numbers (a step that samples its own noise): the generator is put back afterwards,
        # so a run draws the same noise whether its graph was captured now or replayed from a cache -- which is what lets a
        # sampling run that had to be repeated in another arithmetic replay the SAME trajectory
        rng = torch.cuda.get_rng_state()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            fn()
        torch.cuda.set_rng_state(rng)

    def replay(self):
        self.graph.replay()
