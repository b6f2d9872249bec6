#!/usr/bin/env python3
"""scripts/gcn_interp.py -- a small in-order interpreter for the gfx9-family ISA subset the chain kernels compile to (one wave,
64 lanes, numpy), written to look at the round-3 miscompile of k_pair<2, 8, 1> WITHOUT a GPU (NOTES/traps.md): two builds of one
kernel are run on the same random inputs and their outputs compared.  Everything executes in program order and completes at once
(`s_waitcnt`, `s_nop` are no-ops): what it can show is a LOGICAL difference between two listings -- not a missing wait or a
hardware hazard.  It knows only the ~80 opcodes those kernels use and raises on anything else.

    hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S sgm_aggregate.hip -o a.s
    python scripts/gcn_interp.py pair a.s b.s            # k_pair<2, 8, 1>: both listings, chains of 1 .. 4 segments, compare S
"""
from __future__ import annotations

import re
import sys

import numpy as np

M32 = 0xFFFFFFFF
LANES = np.arange(64)


class Unknown(Exception):
    pass


def parse_kernel(path, prefix):
    s = open(path).read().split("\n")
    start = [i for i, l in enumerate(s) if l.startswith(prefix) and ":" in l][0]
    end = [i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end")][0]
    prog, labels = [], {}
    for l in s[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(prog)
            continue
        t = l.split(";")[0].strip()
        if not l.startswith("\t") or not t or t.startswith("."):
            continue
        op, _, rest = t.partition(" ")
        rest = rest.strip()
        # split operands at top-level commas (not inside [...])
        args, depth, cur = [], 0, ""
        for ch in rest:
            if ch == "[":
                depth += 1
            if ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                args.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur.strip())
        # the last operand may carry modifiers separated by spaces
        mods = {}
        if args:
            toks = re.findall(r"[^\s\[]+(?:\[[^\]]*\])?", args[-1])
            args[-1] = toks[0]
            for tk in toks[1:]:
                if ":" in tk:
                    k, v = tk.split(":", 1)
                    mods[k] = v
                else:
                    mods[tk] = True
        prog.append((op, args, mods, t))
    return prog, labels


class Wave:
    def __init__(self, mem: np.ndarray, lds_bytes=65536):
        self.s = np.zeros(128, np.uint64)           # 32-bit values kept in 64-bit slots
        self.v = np.zeros((256, 64), np.uint64)
        self.vcc = 0
        self.exec = (1 << 64) - 1
        self.scc = 0
        self.mem = mem                              # flat byte array = the device's address space (addresses are offsets)
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.trace = None

    # ---- scalar operands
    def rs(self, a):
        if a == "vcc":
            return self.vcc
        if a in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi"):
            whole = self.vcc if a.startswith("vcc") else self.exec
            return (whole >> (32 if a.endswith("hi") else 0)) & M32
        if a == "exec":
            return self.exec
        if a == "scc":
            return self.scc
        m = re.match(r"^s(\d+)$", a)
        if m:
            return int(self.s[int(m.group(1))])
        m = re.match(r"^s\[(\d+):(\d+)\]$", a)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            val = 0
            for k in range(hi, lo - 1, -1):
                val = (val << 32) | int(self.s[k])
            return val
        return self.lit(a)

    @staticmethod
    def lit(a):
        try:
            return int(a, 0)
        except ValueError:
            raise Unknown("operand " + a)

    def ws(self, a, val):
        if a == "vcc":
            self.vcc = val & ((1 << 64) - 1)
            return
        if a in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi"):
            sh = 32 if a.endswith("hi") else 0
            whole = self.vcc if a.startswith("vcc") else self.exec
            whole = (whole & ~(M32 << sh)) | ((val & M32) << sh)
            if a.startswith("vcc"):
                self.vcc = whole
            else:
                self.exec = whole
            return
        if a == "exec":
            self.exec = val & ((1 << 64) - 1)
            return
        m = re.match(r"^s(\d+)$", a)
        if m:
            self.s[int(m.group(1))] = val & M32
            return
        m = re.match(r"^s\[(\d+):(\d+)\]$", a)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            for k in range(lo, hi + 1):
                self.s[k] = val & M32
                val >>= 32
            return
        raise Unknown("scalar destination " + a)

    # ---- vector operands (a lane vector of uint64 holding 32-bit values; 64-bit forms return a Python-int-safe uint64 vector)
    def rv(self, a):
        m = re.match(r"^v(\d+)$", a)
        if m:
            return self.v[int(m.group(1))].copy()
        if re.match(r"^(s\d+|vcc|exec|s\[)", a):   # (vcc_lo / exec_hi too)
            return np.full(64, self.rs(a) & M32, np.uint64)
        return np.full(64, self.lit(a) & M32, np.uint64)

    def rv64(self, a):
        m = re.match(r"^v\[(\d+):(\d+)\]$", a)
        if m:
            lo = int(m.group(1))
            return self.v[lo] | (self.v[lo + 1] << np.uint64(32))
        m = re.match(r"^s\[(\d+):(\d+)\]$", a)
        if m:
            return np.full(64, self.rs(a), np.uint64)
        return np.full(64, self.lit(a) & ((1 << 64) - 1), np.uint64)

    _masks = {}

    def mask(self):
        m = Wave._masks.get(self.exec)
        if m is None:
            m = Wave._masks[self.exec] = np.array([(self.exec >> i) & 1 for i in range(64)], bool)
            m.setflags(write=False)
        return m

    def wv(self, a, val, mask=None):
        if mask is None:
            mask = self.mask()
        m = re.match(r"^v(\d+)$", a)
        if m:
            r = int(m.group(1))
            self.v[r][mask] = (val & np.uint64(M32))[mask]
            return
        m = re.match(r"^v\[(\d+):(\d+)\]$", a)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            if isinstance(val, list):
                for k, part in enumerate(val):
                    self.v[lo + k][mask] = (part & np.uint64(M32))[mask]
            else:
                for k in range(hi - lo + 1):
                    self.v[lo + k][mask] = ((val >> np.uint64(32 * k)) & np.uint64(M32))[mask]
            return
        raise Unknown("vector destination " + a)

    # ---- memory
    def ld(self, addr, nbytes):
        return int.from_bytes(self.mem[addr:addr + nbytes].tobytes(), "little") if addr + nbytes <= len(self.mem) else 0

    def ld32v(self, addrs, ndw, mask):
        out = [np.zeros(64, np.uint64) for _ in range(ndw)]
        if not mask.any():
            return out
        a = addrs[mask].astype(np.int64)
        if a.min() < 0 or a.max() + 4 * ndw > len(self.mem) or (a & 3).any():
            raise Unknown("load outside the simulated memory (or not dword-aligned): 0x%x" % int(a.max()))
        m32 = self.mem.view(np.uint32)
        for k in range(ndw):
            out[k][mask] = m32[(a >> 2) + k]
        return out

    def st32v(self, addrs, vals, mask):
        if not mask.any():
            return
        a = addrs[mask].astype(np.int64)
        if a.min() < 0 or a.max() + 4 * len(vals) > len(self.mem) or (a & 3).any():
            raise Unknown("store outside the simulated memory (or not dword-aligned): 0x%x" % int(a.max()))
        m32 = self.mem.view(np.uint32)
        for k, part in enumerate(vals):
            m32[(a >> 2) + k] = (part[mask] & np.uint64(M32)).astype(np.uint32)


def pk(fn, a, b):
    lo = fn(a & np.uint64(0xFFFF), b & np.uint64(0xFFFF)) & np.uint64(0xFFFF)
    hi = fn((a >> np.uint64(16)) & np.uint64(0xFFFF), (b >> np.uint64(16)) & np.uint64(0xFFFF)) & np.uint64(0xFFFF)
    return lo | (hi << np.uint64(16))


def opsel(w: Wave, args, mods):
    """src0, src1 of a packed op with op_sel_hi applied (default [1,1]: the high half comes from the high half)."""
    a, b = w.rv(args[1]), w.rv(args[2])
    hi = mods.get("op_sel_hi", "[1,1]").strip("[]").split(",")
    if "op_sel" in mods:
        raise Unknown("op_sel")
    if hi[0] == "0":
        a = (a & np.uint64(0xFFFF)) | ((a & np.uint64(0xFFFF)) << np.uint64(16))
    if hi[1] == "0":
        b = (b & np.uint64(0xFFFF)) | ((b & np.uint64(0xFFFF)) << np.uint64(16))
    return a, b


def dpp_src(w: Wave, src, mods):
    """(value per lane, valid per lane) of a DPP source"""
    val = np.zeros(64, np.uint64)
    ok = np.ones(64, bool)
    if "quad_perm" in mods:
        p = [int(x) for x in mods["quad_perm"].strip("[]").split(",")]
        idx = (LANES & ~3) + np.array([p[i & 3] for i in range(64)])
    elif "row_mirror" in mods:
        idx = (LANES & ~15) + (15 - (LANES & 15))
    elif "row_half_mirror" in mods:
        idx = (LANES & ~7) + (7 - (LANES & 7))
    elif "row_bcast" in mods:
        n = int(mods["row_bcast"])
        if n == 15:
            idx = (LANES & ~15) - 1
            ok = LANES >= 16
        else:
            idx = (LANES & ~31) - 1
            ok = LANES >= 32
        idx = np.where(ok, idx, 0)
    elif "wave_shr" in mods:
        idx = LANES - 1
        ok = LANES >= 1
        idx = np.where(ok, idx, 0)
    elif "wave_shl" in mods:
        idx = LANES + 1
        ok = LANES <= 62
        idx = np.where(ok, idx, 0)
    elif "row_shr" in mods:
        n = int(mods["row_shr"])
        ok = (LANES & 15) >= n
        idx = np.where(ok, LANES - n, 0)
    elif "row_shl" in mods:
        n = int(mods["row_shl"])
        ok = (LANES & 15) + n <= 15
        idx = np.where(ok, LANES + n, 0)
    else:
        raise Unknown("dpp control " + str(mods))
    val = src[idx]
    # a source lane that is disabled in EXEC counts as invalid too
    em = w.mask()
    ok = ok & em[idx]
    rm, bm = int(mods.get("row_mask", "0xf"), 0), int(mods.get("bank_mask", "0xf"), 0)
    en = np.array([((rm >> (l >> 4)) & 1) and ((bm >> ((l >> 2) & 3)) & 1) for l in range(64)], bool)
    if mods.get("bound_ctrl") in ("0", "1", True):
        val = np.where(ok, val, 0).astype(np.uint64)
        ok = np.ones(64, bool)
    return val, ok & en


def sat_add16(a, b):
    return np.minimum(a + b, np.uint64(0xFFFF))


def run(prog, labels, w: Wave, max_steps=60_000):
    """one wave to the end (a barrier is a no-op for a wave alone); returns the number of instructions executed"""
    gen = run_gen(prog, labels, w, max_steps)
    while True:
        try:
            next(gen)
        except StopIteration as e:
            return e.value


def run_group(prog, labels, waves, max_steps=200_000):
    """the waves of one workgroup (they share `lds`), switched at every s_barrier; returns the instructions executed per wave"""
    gens = [run_gen(prog, labels, w, max_steps) for w in waves]
    done = [None] * len(waves)
    while any(d is None for d in done):
        for i, gen in enumerate(gens):
            if done[i] is None:
                try:
                    next(gen)                       # runs to its next barrier
                except StopIteration as e:
                    done[i] = e.value
    return done


def run_gen(prog, labels, w: Wave, max_steps=60_000):
    pc, steps = 0, 0
    U64 = np.uint64
    while True:
        steps += 1
        if steps > max_steps:
            raise Unknown("step limit")
        op, a, mods, text = prog[pc]
        pc += 1
        if w.trace is not None:
            w.trace(pc - 1, text)
        # ------------------------------------------------ no-ops / control
        if op == "s_barrier":
            yield "barrier"
            continue
        if op in ("s_nop", "s_waitcnt", "s_setprio", "s_sleep"):
            continue
        if op == "s_endpgm":
            return steps
        if op == "s_branch":
            pc = labels[a[0]]
            continue
        if op.startswith("s_cbranch_"):
            c = op[len("s_cbranch_"):]
            take = {"scc0": w.scc == 0, "scc1": w.scc == 1, "vccz": w.vcc == 0, "vccnz": w.vcc != 0, "execz": w.exec == 0, "execnz": w.exec != 0}[c]
            if take:
                pc = labels[a[0]]
            continue
        # ------------------------------------------------ scalar memory
        if op.startswith("s_load_dword"):
            n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[op]
            base = w.rs(a[1]) + w.rs(a[2]) + int(mods.get("offset", "0"), 0)
            lo = int(re.match(r"^s\[?(\d+)", a[0]).group(1))
            for k in range(n):
                w.s[lo + k] = w.ld(base + 4 * k, 4)
            continue
        # ------------------------------------------------ scalar ALU
        if op.startswith("s_"):
            sfx = op[2:]
            if sfx in ("mov_b32", "mov_b64"):
                w.ws(a[0], w.rs(a[1]))
            elif sfx == "movk_i32":
                k16 = w.lit(a[1]) & 0xFFFF
                w.ws(a[0], (k16 - 0x10000 if k16 & 0x8000 else k16) & M32)
            elif sfx in ("add_i32", "add_u32", "sub_i32", "sub_u32", "addc_u32", "subb_u32", "addk_i32"):
                if sfx == "addk_i32":                # the 16-bit immediate is signed, however it is spelt
                    k16 = w.lit(a[1]) & 0xFFFF
                    x, y = w.rs(a[0]), (k16 - 0x10000 if k16 & 0x8000 else k16)
                else:
                    x, y = w.rs(a[1]), w.rs(a[2])
                x &= M32
                y &= M32
                if sfx in ("add_u32", "addc_u32"):
                    r = x + y + (w.scc if sfx == "addc_u32" else 0)
                    w.scc = int(r > M32)
                elif sfx in ("sub_u32", "subb_u32"):
                    r = x - y - (w.scc if sfx == "subb_u32" else 0)
                    w.scc = int(r < 0)
                else:
                    sx, sy = x - (1 << 32) * (x >> 31), y - (1 << 32) * (y >> 31)
                    r = sx + sy if sfx != "sub_i32" else sx - sy
                    w.scc = int(r > 0x7FFFFFFF or r < -0x80000000)
                w.ws(a[0], r & M32)
            elif sfx in ("mul_i32",):
                w.ws(a[0], (w.rs(a[1]) * w.rs(a[2])) & M32)
            elif sfx == "mulk_i32":
                k16 = w.lit(a[1]) & 0xFFFF
                w.ws(a[0], (w.rs(a[0]) * (k16 - 0x10000 if k16 & 0x8000 else k16)) & M32)
            elif sfx == "mul_hi_u32":
                w.ws(a[0], ((w.rs(a[1]) & M32) * (w.rs(a[2]) & M32)) >> 32)
            elif sfx == "mul_hi_i32":
                x, y = w.rs(a[1]) & M32, w.rs(a[2]) & M32
                x -= (1 << 32) * (x >> 31)
                y -= (1 << 32) * (y >> 31)
                w.ws(a[0], ((x * y) >> 32) & M32)
            elif sfx in ("lshl_b32", "lshr_b32", "ashr_i32", "lshl_b64", "lshr_b64"):
                x, sh = w.rs(a[1]), w.rs(a[2])
                if sfx == "lshl_b32":
                    r = (x << (sh & 31)) & M32
                elif sfx == "lshr_b32":
                    r = (x & M32) >> (sh & 31)
                elif sfx == "ashr_i32":
                    x &= M32
                    x -= (1 << 32) * (x >> 31)
                    r = (x >> (sh & 31)) & M32
                elif sfx == "lshl_b64":
                    r = (x << (sh & 63)) & ((1 << 64) - 1)
                else:
                    r = (x & ((1 << 64) - 1)) >> (sh & 63)
                w.ws(a[0], r)
                w.scc = int(r != 0)
            elif sfx in ("and_b32", "or_b32", "xor_b32", "and_b64", "or_b64", "xor_b64", "andn2_b64", "orn2_b64", "andn2_b32"):
                x, y = w.rs(a[1]), w.rs(a[2])
                full = M32 if sfx.endswith("b32") else (1 << 64) - 1
                x &= full
                y &= full
                r = {"and": x & y, "or": x | y, "xor": x ^ y, "andn2": x & ~y, "orn2": x | (~y & full)}[sfx.split("_")[0]] & full
                w.ws(a[0], r)
                w.scc = int(r != 0)
            elif sfx in ("cselect_b32", "cselect_b64"):
                w.ws(a[0], w.rs(a[1]) if w.scc else w.rs(a[2]))
            elif sfx.startswith("cmpk_"):
                _, cond, ty = sfx.split("_")
                x, k16 = w.rs(a[0]) & M32, w.lit(a[1]) & 0xFFFF
                if ty == "i32":
                    x -= (1 << 32) * (x >> 31)
                    k16 = k16 - 0x10000 if k16 & 0x8000 else k16
                w.scc = int({"eq": x == k16, "lg": x != k16, "gt": x > k16, "ge": x >= k16, "lt": x < k16, "le": x <= k16}[cond])
            elif sfx.startswith("cmp_"):
                _, cond, ty = sfx.split("_")
                x, y = w.rs(a[0]), w.rs(a[1])
                bits = 64 if ty == "u64" else 32
                full = (1 << bits) - 1
                x &= full
                y &= full
                if ty[0] == "i":
                    x -= (1 << bits) * (x >> (bits - 1))
                    y -= (1 << bits) * (y >> (bits - 1))
                w.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[cond])
            elif sfx in ("max_i32", "min_i32", "max_u32", "min_u32"):
                x, y = w.rs(a[1]) & M32, w.rs(a[2]) & M32
                if sfx.endswith("i32"):
                    sx, sy = x - (1 << 32) * (x >> 31), y - (1 << 32) * (y >> 31)
                else:
                    sx, sy = x, y
                first = sx >= sy if sfx.startswith("max") else sx <= sy
                w.ws(a[0], x if first else y)
                w.scc = int(first)
            elif sfx == "pack_ll_b32_b16":
                w.ws(a[0], (w.rs(a[1]) & 0xFFFF) | ((w.rs(a[2]) & 0xFFFF) << 16))
            elif sfx == "bitcmp0_b32":
                w.scc = int(((w.rs(a[0]) >> (w.rs(a[1]) & 31)) & 1) == 0)
            elif sfx == "bitcmp1_b32":
                w.scc = int(((w.rs(a[0]) >> (w.rs(a[1]) & 31)) & 1) == 1)
            elif sfx == "andn2_saveexec_b64":
                old = w.exec
                w.exec = w.rs(a[1]) & ~old & ((1 << 64) - 1)
                w.ws(a[0], old)
                w.scc = int(w.exec != 0)
            elif sfx == "or_saveexec_b64":
                old = w.exec
                w.exec = (w.rs(a[1]) | old) & ((1 << 64) - 1)
                w.ws(a[0], old)
                w.scc = int(w.exec != 0)
            elif sfx == "bcnt1_i32_b64":
                r = bin(w.rs(a[1]) & ((1 << 64) - 1)).count("1")
                w.ws(a[0], r)
                w.scc = int(r != 0)
            elif sfx == "and_saveexec_b64":
                old = w.exec
                w.exec = w.rs(a[1]) & old
                w.ws(a[0], old)
                w.scc = int(w.exec != 0)
            elif sfx == "not_b32":
                r = ~w.rs(a[1]) & M32
                w.ws(a[0], r)
                w.scc = int(r != 0)
            elif sfx == "not_b64":
                r = ~w.rs(a[1]) & ((1 << 64) - 1)
                w.ws(a[0], r)
                w.scc = int(r != 0)
            else:
                raise Unknown(text)
            continue
        # ------------------------------------------------ vector ALU
        em = w.mask()
        if op in ("v_mov_b32_e32", "v_mov_b32_e64"):
            w.wv(a[0], w.rv(a[1]))
        elif op == "v_mov_b64_e32":
            w.wv(a[0], w.rv64(a[1]))
        elif op == "v_mov_b32_dpp":
            val, ok = dpp_src(w, w.rv(a[1]), mods)
            w.wv(a[0], val, em & ok)
        elif op == "v_min_u32_dpp":
            val, ok = dpp_src(w, w.rv(a[1]), mods)
            w.wv(a[0], np.minimum(val, w.rv(a[2])), em & ok)
        elif op in ("v_min_u32_e32", "v_min_u32_e64"):
            w.wv(a[0], np.minimum(w.rv(a[1]), w.rv(a[2])))
        elif op in ("v_pk_min_u16", "v_pk_max_u16", "v_pk_add_u16", "v_pk_sub_i16", "v_pk_sub_u16", "v_pk_add_i16"):
            x, y = opsel(w, a, mods)
            if op == "v_pk_min_u16":
                r = pk(np.minimum, x, y)
            elif op == "v_pk_max_u16":
                r = pk(np.maximum, x, y)
            elif op == "v_pk_add_u16":
                r = pk(sat_add16 if "clamp" in mods else (lambda p, q: p + q), x, y)
            elif op == "v_pk_sub_u16":
                r = pk((lambda p, q: np.where(p >= q, p - q, 0).astype(np.uint64)) if "clamp" in mods else (lambda p, q: (p + np.uint64(0x10000) - q)), x, y)
            elif op == "v_pk_sub_i16":
                if "clamp" in mods:
                    raise Unknown(text)
                r = pk(lambda p, q: (p + np.uint64(0x10000) - q), x, y)
            else:
                if "clamp" in mods:
                    raise Unknown(text)
                r = pk(lambda p, q: p + q, x, y)
            w.wv(a[0], r)
        elif op == "v_alignbit_b32":
            sh = w.rv(a[3]) & U64(31)
            w.wv(a[0], (((w.rv(a[1]) << U64(32)) | w.rv(a[2])) >> sh) & U64(M32))
        elif op == "v_min_u16_sdwa":
            def sel(x, how):
                return {"DWORD": x & U64(0xFFFF), "WORD_0": x & U64(0xFFFF), "WORD_1": (x >> U64(16)) & U64(0xFFFF)}[how]
            if mods.get("dst_sel") != "DWORD" or mods.get("dst_unused") != "UNUSED_PAD":
                raise Unknown(text)
            w.wv(a[0], np.minimum(sel(w.rv(a[1]), mods["src0_sel"]), sel(w.rv(a[2]), mods["src1_sel"])))
        elif op == "v_readlane_b32":
            w.ws(a[0], int(w.rv(a[1])[w.rs(a[2]) & 63]))
        elif op == "v_readfirstlane_b32":
            first = next((l for l in range(64) if em[l]), 0)
            w.ws(a[0], int(w.rv(a[1])[first]))
        elif op in ("v_add_u32_e32", "v_add_u32_e64"):
            w.wv(a[0], (w.rv(a[1]) + w.rv(a[2])) & U64(M32))
        elif op in ("v_subrev_u32_e32", "v_subrev_u32_e64"):
            w.wv(a[0], (w.rv(a[2]) + U64(1 << 32) - w.rv(a[1])) & U64(M32))
        elif op in ("v_ashrrev_i32_e32", "v_ashrrev_i32_e64"):
            x = w.rv(a[2]).astype(np.int64)
            x = np.where(x >> 31, x - (1 << 32), x)
            w.wv(a[0], ((x >> (w.rv(a[1]) & U64(31)).astype(np.int64)) & M32).astype(np.uint64))
        elif op in ("v_sub_u32_e32", "v_sub_u32_e64"):
            w.wv(a[0], (w.rv(a[1]) + U64(1 << 32) - w.rv(a[2])) & U64(M32))
        elif op in ("v_lshlrev_b32_e32", "v_lshlrev_b32_e64"):
            w.wv(a[0], (w.rv(a[2]) << (w.rv(a[1]) & U64(31))) & U64(M32))
        elif op in ("v_lshrrev_b32_e32", "v_lshrrev_b32_e64"):
            w.wv(a[0], (w.rv(a[2]) >> (w.rv(a[1]) & U64(31))) & U64(M32))
        elif op in ("v_and_b32_e32", "v_and_b32_e64"):
            w.wv(a[0], w.rv(a[1]) & w.rv(a[2]))
        elif op in ("v_or_b32_e32", "v_or_b32_e64"):
            w.wv(a[0], w.rv(a[1]) | w.rv(a[2]))
        elif op == "v_lshl_or_b32":
            w.wv(a[0], ((w.rv(a[1]) << (w.rv(a[2]) & U64(31))) | w.rv(a[3])) & U64(M32))
        elif op == "v_lshl_add_u32":
            w.wv(a[0], ((w.rv(a[1]) << (w.rv(a[2]) & U64(31))) + w.rv(a[3])) & U64(M32))
        elif op == "v_add3_u32":
            w.wv(a[0], (w.rv(a[1]) + w.rv(a[2]) + w.rv(a[3])) & U64(M32))
        elif op in ("v_min3_u32",):
            w.wv(a[0], np.minimum(np.minimum(w.rv(a[1]), w.rv(a[2])), w.rv(a[3])))
        elif op == "v_or3_b32":
            w.wv(a[0], w.rv(a[1]) | w.rv(a[2]) | w.rv(a[3]))
        elif op == "v_and_or_b32":
            w.wv(a[0], (w.rv(a[1]) & w.rv(a[2])) | w.rv(a[3]))
        elif op == "v_bfi_b32":
            m_ = w.rv(a[1])
            w.wv(a[0], (m_ & w.rv(a[2])) | (~m_ & U64(M32) & w.rv(a[3])))
        elif op in ("v_xor_b32_e32", "v_xor_b32_e64"):
            w.wv(a[0], w.rv(a[1]) ^ w.rv(a[2]))
        elif op in ("v_mul_lo_u32",):
            w.wv(a[0], (w.rv(a[1]) * w.rv(a[2])) & U64(M32))
        elif op == "v_mul_hi_u32":
            w.wv(a[0], np.array([(int(x) * int(y)) >> 32 for x, y in zip(w.rv(a[1]), w.rv(a[2]))], np.uint64))
        elif op in ("v_mul_u32_u24_e32", "v_mul_u32_u24_e64", "v_mul_u32_u24_sdwa"):
            x, y = w.rv(a[1]), w.rv(a[2])
            if op.endswith("sdwa"):
                if mods.get("dst_sel") != "DWORD" or mods.get("dst_unused") != "UNUSED_PAD":
                    raise Unknown(text)
                pick = {"DWORD": lambda q: q, "WORD_0": lambda q: q & U64(0xFFFF), "WORD_1": lambda q: (q >> U64(16)) & U64(0xFFFF),
                        "BYTE_0": lambda q: q & U64(0xFF), "BYTE_1": lambda q: (q >> U64(8)) & U64(0xFF), "BYTE_2": lambda q: (q >> U64(16)) & U64(0xFF),
                        "BYTE_3": lambda q: (q >> U64(24)) & U64(0xFF)}
                x, y = pick[mods["src0_sel"]](x), pick[mods["src1_sel"]](y)
            w.wv(a[0], ((x & U64(0xFFFFFF)) * (y & U64(0xFFFFFF))) & U64(M32))
        elif op in ("v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_add_co_u32_e32", "v_add_co_u32_e64", "v_sub_co_u32_e32", "v_sub_co_u32_e64",
                    "v_subb_co_u32_e32", "v_subb_co_u32_e64"):
            x, y = w.rv(a[2]).astype(object), w.rv(a[3]).astype(object)
            cin = np.zeros(64, object)
            if "addc" in op or "subb" in op:
                cm = w.rs(a[4])
                cin = np.array([(cm >> l) & 1 for l in range(64)], object)
            r = x - y - cin if "sub" in op else x + y + cin
            carry = 0
            for l in range(64):
                if em[l] and (r[l] < 0 or r[l] > M32):
                    carry |= 1 << l
            w.wv(a[0], np.array([int(q) & M32 for q in r], np.uint64))
            w.ws(a[1], carry)
        elif op in ("v_cvt_f32_u32_e32", "v_cvt_f32_u32_e64"):
            w.wv(a[0], w.rv(a[1]).astype(np.float32).view(np.uint32).astype(np.uint64))
        elif op in ("v_cvt_i32_f32_e32", "v_cvt_u32_f32_e32"):
            f = w.rv(a[1]).astype(np.uint32).view(np.float32).astype(np.float64)
            lo_, hi_ = (-2147483648.0, 2147483647.0) if "i32" in op else (0.0, 4294967295.0)
            t_ = np.clip(np.trunc(np.nan_to_num(f)), lo_, hi_).astype(np.int64)
            w.wv(a[0], (t_ & M32).astype(np.uint64))
        elif op in ("v_mul_f32_e32", "v_mul_f32_e64"):
            f1 = w.rv(a[1]).astype(np.uint32).view(np.float32)
            f2 = w.rv(a[2]).astype(np.uint32).view(np.float32)
            w.wv(a[0], (f1 * f2).astype(np.float32).view(np.uint32).astype(np.uint64))
        elif op in ("v_rcp_iflag_f32_e32", "v_rcp_f32_e32"):
            f1 = w.rv(a[1]).astype(np.uint32).view(np.float32)
            with np.errstate(divide="ignore", over="ignore"):
                w.wv(a[0], (np.float32(1.0) / f1).astype(np.float32).view(np.uint32).astype(np.uint64))
        elif op == "v_perm_b32":
            s0, s1, sel = w.rv(a[1]), w.rv(a[2]), w.rv(a[3])
            both = s1 | (s0 << U64(32))
            out = np.zeros(64, np.uint64)
            for k in range(4):
                sb = (sel >> U64(8 * k)) & U64(0xFF)
                if ((sb >= 8) & (sb < 12)).any():
                    raise Unknown("v_perm_b32 sign selectors")
                byte = np.where(sb < 8, (both >> (np.minimum(sb, 7) * U64(8))) & U64(0xFF), np.where(sb == 12, 0, 0xFF)).astype(np.uint64)
                out |= byte << U64(8 * k)
            w.wv(a[0], out)
        elif op == "v_writelane_b32":
            w.v[int(a[0][1:])][w.rs(a[2]) & 63] = w.rs(a[1]) & M32
        elif op == "v_bfe_u32":
            off_, wid = w.rv(a[2]) & U64(31), w.rv(a[3]) & U64(31)
            w.wv(a[0], (w.rv(a[1]) >> off_) & ((U64(1) << wid) - U64(1)))
        elif op in ("v_max_u32_e32", "v_max_u32_e64"):
            w.wv(a[0], np.maximum(w.rv(a[1]), w.rv(a[2])))
        elif op == "v_mad_u64_u32":                 # v[d:d+1], carry-out pair, a, b, c64
            w.wv(a[0], ((w.rv(a[2]) * w.rv(a[3])) + w.rv64(a[4])))
        elif op == "v_lshl_add_u64":
            w.wv(a[0], (w.rv64(a[1]) << (w.rv(a[2]) & U64(63))) + w.rv64(a[3]))
        elif op == "v_min_i32_e32" or op == "v_min_i32_e64" or op == "v_max_i32_e32" or op == "v_max_i32_e64":
            x, y = w.rv(a[1]).astype(np.int64), w.rv(a[2]).astype(np.int64)
            x = np.where(x >> 31, x - (1 << 32), x)
            y = np.where(y >> 31, y - (1 << 32), y)
            r = np.minimum(x, y) if "min" in op else np.maximum(x, y)
            w.wv(a[0], (r & M32).astype(np.uint64))
        elif re.match(r"^v_cmp_(ne|eq|lt|gt|le|ge)_(u32|i32|i64|u64)_e(32|64)$", op):
            srcs = a[1:]                              # (the e32 form spells its destination too: "v_cmp_gt_u32_e32 vcc, 5, v50")
            ty = op.split("_")[3]
            if ty in ("i64", "u64"):
                x, y = w.rv64(srcs[0]), w.rv64(srcs[1])
                if ty == "i64":
                    x, y = x.astype(np.int64), y.astype(np.int64)
            else:
                x, y = w.rv(srcs[0]).astype(np.int64), w.rv(srcs[1]).astype(np.int64)
                if ty == "i32":
                    x = np.where(x >> 31, x - (1 << 32), x)
                    y = np.where(y >> 31, y - (1 << 32), y)
            cond = op.split("_")[2]
            r = {"ne": x != y, "eq": x == y, "lt": x < y, "gt": x > y, "le": x <= y, "ge": x >= y}[cond]
            bits = 0
            for l in range(64):
                if em[l] and r[l]:
                    bits |= 1 << l
            w.ws(a[0], bits)
        elif op == "v_cndmask_b32_e64":
            sel = w.rs(a[3])
            pick = np.array([(sel >> l) & 1 for l in range(64)], bool)
            w.wv(a[0], np.where(pick, w.rv(a[2]), w.rv(a[1])))
        elif op == "v_cndmask_b32_e32":
            pick = np.array([(w.vcc >> l) & 1 for l in range(64)], bool)
            w.wv(a[0], np.where(pick, w.rv(a[2]), w.rv(a[1])))
        # ------------------------------------------------ LDS
        elif op in ("ds_write2st64_b64", "ds_read2st64_b64", "ds_read_b64", "ds_write_b64", "ds_read_b128", "ds_write_b128", "ds_read2_b64", "ds_write2_b64"):
            base = w.rv(a[0] if op.startswith("ds_write") else a[1])
            def lds_rw(off, regs, write, ndw):
                ad = base.astype(np.int64) + off
                ok = em & (ad + 4 * ndw <= len(w.lds))          # beyond the allocation: the hardware drops the write and returns zeros
                if (ad[ok] & 3).any():
                    raise Unknown("LDS access not dword-aligned")
                l32 = w.lds.view(np.uint32)
                for k in range(ndw):
                    if write:
                        l32[(ad[ok] >> 2) + k] = w.v[regs + k][ok].astype(np.uint32)
                    else:
                        w.v[regs + k][ok] = l32[(ad[ok] >> 2) + k]
                        w.v[regs + k][em & ~ok] = 0
            def lo_of(x):
                return int(re.match(r"^v\[?(\d+)", x).group(1))
            if op in ("ds_write2st64_b64", "ds_write2_b64"):
                unit = 512 if "st64" in op else 8
                lds_rw(int(mods.get("offset0", "0")) * unit, lo_of(a[1]), True, 2)
                lds_rw(int(mods.get("offset1", "0")) * unit, lo_of(a[2]), True, 2)
            elif op in ("ds_read2st64_b64", "ds_read2_b64"):
                unit = 512 if "st64" in op else 8
                lds_rw(int(mods.get("offset0", "0")) * unit, lo_of(a[0]), False, 2)
                lds_rw(int(mods.get("offset1", "0")) * unit, lo_of(a[0]) + 2, False, 2)
            elif op == "ds_read_b64":
                lds_rw(int(mods.get("offset", "0")), lo_of(a[0]), False, 2)
            elif op == "ds_write_b64":
                lds_rw(int(mods.get("offset", "0")), lo_of(a[1]), True, 2)
            elif op == "ds_read_b128":
                lds_rw(int(mods.get("offset", "0")), lo_of(a[0]), False, 4)
            else:
                lds_rw(int(mods.get("offset", "0")), lo_of(a[1]), True, 4)
        # ------------------------------------------------ buffer / global
        elif op.startswith(("buffer_load_dword", "buffer_store_dword")):
            ndw = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}[op.split("_")[2]]
            rs = re.match(r"^s\[(\d+):(\d+)\]$", a[2])
            lo = int(rs.group(1))
            base = int(w.s[lo]) | ((int(w.s[lo + 1]) & 0xFFFF) << 32)
            nrec = int(w.s[lo + 2])
            if (int(w.s[lo + 1]) >> 16) & 0x3FFF:
                raise Unknown("strided buffer")
            off = (w.rv(a[1]) if "offen" in mods else np.zeros(64, np.uint64)) + U64(int(mods.get("offset", "0"), 0))
            soff = w.rs(a[3]) & M32
            inb = (off + U64(4 * ndw)) <= U64(nrec)          # raw buffer: range check on the offset without soffset
            addrs = (U64(base) + off + U64(soff))
            lo_v = int(re.match(r"^v\[?(\d+)", a[0]).group(1))
            if op.startswith("buffer_load"):
                vals = w.ld32v(addrs, ndw, em & inb)
                for k in range(ndw):
                    w.v[lo_v + k][em & inb] = vals[k][em & inb]
                    w.v[lo_v + k][em & ~inb] = 0
            else:
                w.st32v(addrs, [w.v[lo_v + k] for k in range(ndw)], em & inb)
        elif op.startswith(("scratch_load_dword", "scratch_store_dword")):
            ndw = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}[op.split("_")[2]]
            load = op.startswith("scratch_load")
            vaddr, saddr = (a[1], a[2]) if load else (a[0], a[2])
            data = a[0] if load else a[1]
            off = int(mods.get("offset", "0"), 0) + (0 if saddr == "off" else w.rs(saddr))
            if vaddr != "off":
                raise Unknown("scratch with a vector address: " + text)
            lo_v = int(re.match(r"^v\[?(\d+)", data).group(1))
            if not hasattr(w, "scratch"):
                w.scratch = np.zeros((64, 4096), np.uint32)
            for k in range(ndw):
                if load:
                    w.v[lo_v + k][em] = w.scratch[:, off // 4 + k][em]
                else:
                    w.scratch[:, off // 4 + k][em] = w.v[lo_v + k][em].astype(np.uint32)
        elif op == "global_store_short":
            addrs = w.rv64(a[0]) if a[2] == "off" else U64(w.rs(a[2])) + w.rv(a[0])
            addrs = addrs + U64(int(mods.get("offset", "0"), 0) & ((1 << 64) - 1))
            data = w.rv(a[1])
            for l in range(64):
                if em[l]:
                    ad = int(addrs[l])
                    if ad < 0 or ad + 2 > len(w.mem):
                        raise Unknown("store outside the simulated memory: 0x%x" % ad)
                    w.mem[ad:ad + 2] = np.frombuffer(np.uint16(int(data[l]) & 0xFFFF).tobytes(), np.uint8)
        elif op.startswith(("global_load_dword", "global_store_dword")):
            ndw = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}[op.split("_")[2]]
            load = op.startswith("global_load")
            vaddr, saddr = (a[1], a[2]) if load else (a[0], a[2])
            data = a[0] if load else a[1]
            if saddr == "off":
                addrs = w.rv64(vaddr)
            else:
                addrs = U64(w.rs(saddr)) + w.rv(vaddr)
            addrs = addrs + U64(int(mods.get("offset", "0"), 0) & ((1 << 64) - 1))
            lo_v = int(re.match(r"^v\[?(\d+)", data).group(1))
            if load:
                vals = w.ld32v(addrs, ndw, em)
                for k in range(ndw):
                    w.v[lo_v + k][em] = vals[k][em]
            else:
                w.st32v(addrs, [w.v[lo_v + k] for k in range(ndw)], em)
        else:
            raise Unknown(text)


# ------------------------------------------------------------------------------------------------------------ harnesses
def kernel_case(listing, prefix, buffers, args, chain, seed, outputs):
    """One wave (wave `chain % 4` of workgroup `chain // 4`) of a chain kernel.  buffers: name -> (bytes, upper bound of the random u16
    fill or None for zeros); args: list of ("ptr", name | None) / ("i32", value); outputs: buffer names returned (concatenated, as uint32)."""
    prog, labels = parse_kernel(listing, prefix)
    rng = np.random.default_rng(seed)
    offs, total = {"args": 1 << 16}, (1 << 16) + 4096
    for k, (n, _) in buffers.items():
        offs[k] = total
        total += (n + 4095) & ~4095
    mem = np.zeros(total + 4096, np.uint8)
    for k, (n, hi) in buffers.items():
        if hi:
            mem[offs[k]:offs[k] + n] = np.frombuffer(rng.integers(0, hi, n // 2, dtype=np.uint16).tobytes(), np.uint8)
    at = 0
    for kind, val in args:
        size = 8 if kind == "ptr" else 4
        at = (at + size - 1) & ~(size - 1)
        v = (offs[val] if val is not None else 0) if kind == "ptr" else (val & M32)
        mem[offs["args"] + at:offs["args"] + at + size] = np.frombuffer(int(v).to_bytes(size, "little"), np.uint8)
        at += size
    w = Wave(mem)
    w.s[0], w.s[1] = offs["args"] & M32, offs["args"] >> 32
    w.s[2] = chain // 4
    w.v[0] = (chain % 4) * 64 + LANES
    steps = run(prog, labels, w)
    return np.concatenate([np.frombuffer(mem[offs[k]:offs[k] + buffers[k][0]].tobytes(), np.uint32) for k in outputs]), steps


def _sizes(width1, h, NP, K):
    VB = 256 * NP
    npx = width1 * h
    maxseg = (max(width1, h) + K - 1) // K + 1
    nchains = 4 * (width1 + h)
    return VB, npx, maxseg, nchains


def ckpt_case(listing, prefix, width1, h, dx, dy, chain, seed, with_endstate=False, NP=2, K=8):
    """k_ckpt<NP, K>: the forward sweep of a family -- checkpoints, minima, end states"""
    VB, npx, maxseg, nchains = _sizes(width1, h, NP, K)
    buffers = {"C": (npx * VB, 2000), "ckpt": (nchains * maxseg * VB, None), "mins": (nchains * maxseg * K * 2, None), "end": (nchains * VB, None)}
    args = [("ptr", "C"), ("ptr", "ckpt"), ("ptr", "mins")] + [("i32", v) for v in (width1, h, dx, dy, 7, 150, nchains, maxseg)] + [("ptr", "end" if with_endstate else None)]
    return kernel_case(listing, prefix, buffers, args, chain, seed, ("ckpt", "mins", "end"))


def sweep_case(listing, prefix, width1, h, dx, dy, chain, seed, NP=2, K=8):
    """k_sweep<NP, SMODE, U>, SMODE 0 / 1: one unpaired path of the 5-path mode"""
    VB, npx, maxseg, nchains = _sizes(width1, h, NP, K)
    buffers = {"C": (npx * VB, 2000), "S": (npx * VB, 3000), "sel16": (npx * 2, None), "selkey": (npx * 4, None)}
    args = [("ptr", "C"), ("ptr", "S")] + [("i32", v) for v in (width1, h, dx, dy, 7, 150, nchains, 256, 0, 10, 1)] + [("ptr", "sel16"), ("ptr", "selkey")]
    return kernel_case(listing, prefix, buffers, args, chain, seed, ("S",))


def rowsweep_case(listing, prefix, width1, h, row, seed, NP=2, XB=10):
    """k_rowsweep<NP>: paths 0 and 4 over one row -- entry states per block of XB columns, minima after every step"""
    VB = 256 * NP
    nbx = (width1 + XB - 1) // XB
    buffers = {"C": (width1 * h * VB, 2000), "entF": (h * nbx * VB, None), "entB": (h * nbx * VB, None), "MF": (h * width1 * 2 + 64, None), "MB": (h * width1 * 2 + 64, None)}
    args = [("ptr", k) for k in ("C", "entF", "entB", "MF", "MB")] + [("i32", v) for v in (width1, h, 7, 150, nbx)]
    return kernel_case(listing, prefix, buffers, args, row, seed, ("entF", "entB", "MF", "MB"))


def pairx_case(listing, prefix, width1, h, block, seed, NP=2, K=8, XB=10):
    """k_pairx<NP, K, ACC = true, ONE = false>: the column family with the rows riding along -- one workgroup of XB waves (a block of XB
    columns, upper or lower half of the image), barriers included.  Random inputs: states, minima and S are whatever they are."""
    prog, labels = parse_kernel(listing, prefix)
    rng = np.random.default_rng(seed)
    VB = 256 * NP
    npx = width1 * h
    nbx = (width1 + XB - 1) // XB
    maxseg = (h + K - 1) // K + 2
    ncol = 2 * width1 + 16
    buffers = {"C": (npx * VB, 2000), "S": (npx * VB, 3000), "ckpt": (ncol * maxseg * VB, 1500), "mins": (ncol * maxseg * K * 2, 1500),
               "entF": (h * nbx * VB, 1500), "entB": (h * nbx * VB, 1500), "MF": (npx * 2 + 64, 1500), "MB": (npx * 2 + 64, 1500), "end": (ncol * VB, 1500)}
    offs, total = {"args": 1 << 16}, (1 << 16) + 4096
    for k, (n, _) in buffers.items():
        offs[k] = total
        total += (n + 4095) & ~4095
    mem = np.zeros(total + 4096, np.uint8)
    for k, (n, hi) in buffers.items():
        mem[offs[k]:offs[k] + n] = np.frombuffer(rng.integers(0, hi, n // 2, dtype=np.uint16).tobytes(), np.uint8)
    def put(o, val, n):
        mem[offs["args"] + o:offs["args"] + o + n] = np.frombuffer(int(val & ((1 << (8 * n)) - 1)).to_bytes(n, "little"), np.uint8)
    for i, k in enumerate(("C", "S", "ckpt", "mins", "entF", "entB", "MF", "MB")):
        put(8 * i, offs[k], 8)
    put(64, nbx, 4)                             # RowSide.nbx (+ 4 bytes of padding)
    for i, v in enumerate((width1, h, 7, 150, maxseg)):
        put(72 + 4 * i, v, 4)
    put(96, offs["end"], 8)
    lds = np.zeros(160 * 1024, np.uint8)
    waves = []
    for wv in range(XB):
        w = Wave(mem)
        w.lds = lds
        w.s[0], w.s[1] = offs["args"] & M32, offs["args"] >> 32
        w.s[2] = block
        w.v[0] = wv * 64 + LANES
        waves.append(w)
    steps = run_group(prog, labels, waves)
    return np.frombuffer(mem[offs["S"]:offs["S"] + buffers["S"][0]].tobytes(), np.uint32).copy(), steps


def family_case(listing, width1, h, dx, dy, seed, NP=2, K=8, P1=7, P2=150, cmax=2000, smode=0, smax=3000):
    """A whole diagonal family of a small image through the two kernels that compute it -- k_ckpt<NP, K> (forward sweep: checkpoints,
    minima) and then k_pair<NP, K, 0> (S = L_forward + L_backward) -- every chain, one wave at a time, on one memory image.
    Returns (C as u16 [h][width1][128 * NP], S likewise)."""
    ck = parse_kernel(listing, "_ZN4wass6k_ckptILi%dELi%dEEE" % (NP, K))
    pr = parse_kernel(listing, "_ZN4wass6k_pairILi%dELi%dELi%dEEE" % (NP, K, smode))   # 1: S += ...; 2: the last family (S finished, selection; keepS stores it)
    rng = np.random.default_rng(seed)
    VB, npx, maxseg, _ = _sizes(width1, h, NP, K)
    nchains = width1 + h - 1
    buffers = {"C": npx * VB, "S": npx * VB, "ckpt": (nchains + 4) * maxseg * VB, "mins": (nchains + 4) * maxseg * K * 2 + 64, "sel16": npx * 2, "selkey": npx * 4}
    offs, total = {"args": 1 << 16}, (1 << 16) + 4096
    for k, n in buffers.items():
        offs[k] = total
        total += (n + 4095) & ~4095
    mem = np.zeros(total + (1 << 20), np.uint8)    # (the pair kernel re-reads up to K vectors past a chain's end: harmless, but it must be memory)
    mem[offs["C"]:offs["C"] + buffers["C"]] = np.frombuffer(rng.integers(0, cmax, buffers["C"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    mem[offs["S"]:offs["S"] + buffers["S"]] = 0xEE                     # SMODE 0 writes S, it does not read it
    if smode != 0:
        mem[offs["S"]:offs["S"] + buffers["S"]] = np.frombuffer(rng.integers(0, smax, buffers["S"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    s_before = np.frombuffer(mem[offs["S"]:offs["S"] + buffers["S"]].tobytes(), np.uint16).copy()
    def launch(kernel, args):
        at = 0
        for kind, val in args:
            size = 8 if kind == "ptr" else 4
            at = (at + size - 1) & ~(size - 1)
            v = (offs[val] if val is not None else 0) if kind == "ptr" else (val & M32)
            mem[offs["args"] + at:offs["args"] + at + size] = np.frombuffer(int(v).to_bytes(size, "little"), np.uint8)
            at += size
        for c in range(nchains):
            w = Wave(mem)
            w.s[0], w.s[1] = offs["args"] & M32, offs["args"] >> 32
            w.s[2] = c // 4
            w.v[0] = (c % 4) * 64 + LANES
            run(kernel[0], kernel[1], w)
    ints = [("i32", v) for v in (width1, h, dx, dy, P1, P2, nchains, maxseg)]
    launch(ck, [("ptr", "C"), ("ptr", "ckpt"), ("ptr", "mins")] + ints + [("ptr", None)])
    launch(pr, [("ptr", "C"), ("ptr", "S"), ("ptr", "ckpt"), ("ptr", "mins")] + ints + [("i32", v) for v in (128 * NP, 0, 10, 1)] +
           [("ptr", "sel16"), ("ptr", "selkey"), ("ptr", None)])
    shape = (h, width1, 128 * NP)
    out = (np.frombuffer(mem[offs["C"]:offs["C"] + buffers["C"]].tobytes(), np.uint16).reshape(shape).copy(),
           np.frombuffer(mem[offs["S"]:offs["S"] + buffers["S"]].tobytes(), np.uint16).reshape(shape).copy())
    return out if smode == 0 else out + (s_before.reshape(shape),)


class Device:
    """a memory image with named buffers and kernel launches on it (one wave per 64 threads; the waves of a workgroup share LDS)"""
    def __init__(self, buffers, pad=1 << 20):
        self.offs, total = {"args": 1 << 16}, (1 << 16) + 4096
        self.size = dict(buffers)
        for k, n in buffers.items():
            self.offs[k] = total
            total += (n + 4095) & ~4095
        self.mem = np.zeros(total + pad, np.uint8)

    def fill_u16(self, name, rng, hi):
        n = self.size[name]
        self.mem[self.offs[name]:self.offs[name] + n] = np.frombuffer(rng.integers(0, hi, n // 2, dtype=np.uint16).tobytes(), np.uint8)

    def u16(self, name):
        return np.frombuffer(self.mem[self.offs[name]:self.offs[name] + self.size[name]].tobytes(), np.uint16).copy()

    def launch(self, kernel, grid, waves_per_group, args, lds_bytes=65536):
        """args: ("ptr", buffer name | None | (name, byte offset)) / ("i32", value) / ("pad", bytes), laid out with natural alignment"""
        at = 0
        a0 = self.offs["args"]
        self.mem[a0:a0 + 4096] = 0
        for kind, val in args:
            if kind == "pad":
                at += val
                continue
            size = 8 if kind == "ptr" else 4
            at = (at + size - 1) & ~(size - 1)
            if kind == "ptr":
                v = 0 if val is None else (self.offs[val[0]] + val[1] if isinstance(val, tuple) else self.offs[val])
            else:
                v = val & M32
            self.mem[a0 + at:a0 + at + size] = np.frombuffer(int(v).to_bytes(size, "little"), np.uint8)
            at += size
        for g_ in range(grid):
            lds = np.zeros(lds_bytes, np.uint8)
            waves = []
            for wv in range(waves_per_group):
                w = Wave(self.mem)
                w.lds = lds
                w.s[0], w.s[1] = a0 & M32, a0 >> 32
                w.s[2] = g_
                w.v[0] = wv * 64 + LANES
                waves.append(w)
            if waves_per_group == 1 or not any(p[0] == "s_barrier" for p in kernel[0]):
                for w in waves:
                    run(kernel[0], kernel[1], w)
            else:
                run_group(kernel[0], kernel[1], waves)


def single_path_case(listing, width1, h, dx, dy, seed, smode=1, NP=2, U=8, P1=7, P2=150, cmax=2000, smax=3000):
    """k_sweep<NP, SMODE, U> over every chain of a small image: one unpaired path of the 5-path mode (S = / += L_r; 2: the last one, finished and
    stored for the debug fetch).  Returns (C, S before, S after) as u16 [h][width1][128 * NP]."""
    k = parse_kernel(listing, "_ZN4wass7k_sweepILi%dELi%dELi%dEEE" % (NP, smode, U))
    rng = np.random.default_rng(seed)
    VB, npx = 256 * NP, width1 * h
    nch = h if dy == 0 else (width1 if dx == 0 else width1 + h - 1)
    dev = Device({"C": npx * VB, "S": npx * VB, "sel16": npx * 2 + 256, "selkey": npx * 4 + 256})
    dev.fill_u16("C", rng, cmax)
    dev.fill_u16("S", rng, smax)
    s_before = dev.u16("S")
    dev.launch(k, (nch + 3) // 4, 4, [("ptr", "C"), ("ptr", "S")] + [("i32", v) for v in (width1, h, dx, dy, P1, P2, nch, 128 * NP, 0, 10, 1)] +
               [("ptr", "sel16"), ("ptr", "selkey")])
    shape = (h, width1, 128 * NP)
    return dev.u16("C").reshape(shape), s_before.reshape(shape), dev.u16("S").reshape(shape)


def columns_rows_case(listing, width1, h, seed, NP=2, K=8, XB=10, P1=7, P2=150, cmax=2000, smax=3000, one=False):
    """The fused half of the 8-path schedule on a small image, as the host launches it: k_rowsweep<NP> (paths 0 and 4: entry states, minima),
    k_ckpt<NP, K> over the split column family, k_pairx<NP, K, ACC = true, ONE = false> (S += L_2 + L_6 + L_0 + L_4), every workgroup.
    one: the 5-path form k_pairx<NP, K, true, ONE = true> (S += L_2 + L_0 + L_4: the upward column path is computed and left out).
    Returns (C, S before, S after) as u16 [h][width1][128 * NP]."""
    rs_k = parse_kernel(listing, "_ZN4wass10k_rowsweepILi%dEEE" % NP)
    ck_k = parse_kernel(listing, "_ZN4wass6k_ckptILi%dELi%dEEE" % (NP, K))
    px_k = parse_kernel(listing, "_ZN4wass7k_pairxILi%dELi%dELb1ELb%dEEE" % (NP, K, 1 if one else 0))
    rng = np.random.default_rng(seed)
    VB, vec = 256 * NP, 64 * NP
    npx, nbx = width1 * h, (width1 + XB - 1) // XB
    nch = 2 * width1
    mseg = ((h - h // 2) + K - 1) // K
    dev = Device({"C": npx * VB, "S": npx * VB, "ckpt": nch * (mseg + 1) * VB, "mins": nch * mseg * K * 2 + 256, "entF": h * nbx * VB, "entB": h * nbx * VB,
                  "MF": h * nbx * XB * 2 + 256, "MB": h * nbx * XB * 2 + 256})
    dev.fill_u16("C", rng, cmax)
    dev.fill_u16("S", rng, smax)
    s_before = dev.u16("S")
    dev.launch(rs_k, (h + 3) // 4, 4, [("ptr", k) for k in ("C", "entF", "entB", "MF", "MB")] + [("i32", v) for v in (width1, h, P1, P2, nbx)])
    end = ("ckpt", nch * mseg * vec * 4)
    dev.launch(ck_k, (nch + 3) // 4, 4, [("ptr", "C"), ("ptr", "ckpt"), ("ptr", "mins")] + [("i32", v) for v in (width1, h, 0, 1, P1, P2, nch, mseg)] + [("ptr", end)])
    dev.launch(px_k, 2 * nbx, XB, [("ptr", k) for k in ("C", "S", "ckpt", "mins", "entF", "entB", "MF", "MB")] + [("i32", nbx), ("pad", 4)] +
               [("i32", v) for v in (width1, h, P1, P2, mseg)] + [("ptr", end)], lds_bytes=XB * 2 * K * vec * 4)
    shape = (h, width1, 128 * NP)
    return dev.u16("C").reshape(shape), s_before.reshape(shape), dev.u16("S").reshape(shape)


# ------------------------------------------------------------------------------------------------------------ the k_pair experiment
def pair_case(listing, prefix, width1, h, dx, dy, chain, seed, smode_has_S=True, with_endstate=False, NP=2, K=8, whole=False):
    """one wave of k_pair<NP, K, 1> on random inputs; returns the S volume afterwards (uint32 view)"""
    prog, labels = parse_kernel(listing, prefix)
    rng = np.random.default_rng(seed)
    VB = 256 * NP
    npx = width1 * h
    maxseg = (max(width1, h) + K - 1) // K + 1
    nchains = 4 * (width1 + h)
    sizes = {"args": 4096, "C": npx * VB, "S": npx * VB, "ckpt": nchains * maxseg * VB, "mins": nchains * maxseg * K * 2, "sel16": npx * 2, "selkey": npx * 4,
             "end": nchains * VB}
    offs, total = {}, 1 << 16
    for k, n in sizes.items():
        offs[k] = total
        total += (n + 4095) & ~4095
    mem = np.zeros(total + 4096, np.uint8)
    # costs small enough that nothing saturates differently; any values serve for a comparison of two listings
    mem[offs["C"]:offs["C"] + sizes["C"]] = np.frombuffer(rng.integers(0, 2000, sizes["C"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    mem[offs["S"]:offs["S"] + sizes["S"]] = np.frombuffer(rng.integers(0, 3000, sizes["S"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    mem[offs["ckpt"]:offs["ckpt"] + sizes["ckpt"]] = np.frombuffer(rng.integers(0, 1500, sizes["ckpt"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    mem[offs["mins"]:offs["mins"] + sizes["mins"]] = np.frombuffer(rng.integers(0, 1500, sizes["mins"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    mem[offs["end"]:offs["end"] + sizes["end"]] = np.frombuffer(rng.integers(0, 1500, sizes["end"] // 2, dtype=np.uint16).tobytes(), np.uint8)
    P1, P2 = 7, 150
    args = np.zeros(0x68, np.uint8)
    def put(o, val, n):
        args[o:o + n] = np.frombuffer(int(val).to_bytes(n, "little"), np.uint8)
    put(0x00, offs["C"], 8); put(0x08, offs["S"], 8); put(0x10, offs["ckpt"], 8); put(0x18, offs["mins"], 8)
    for i, val in enumerate((width1, h, dx & M32, dy & M32, P1, P2, nchains, maxseg, 256, 0, 10, 1)):
        put(0x20 + 4 * i, val, 4)
    put(0x50, offs["sel16"], 8); put(0x58, offs["selkey"], 8); put(0x60, offs["end"] if with_endstate else 0, 8)
    mem[offs["args"]:offs["args"] + len(args)] = args
    w = Wave(mem)
    w.s[0], w.s[1] = offs["args"] & M32, offs["args"] >> 32
    w.s[2] = chain // 4                         # workgroup id
    w.v[0] = (chain % 4) * 64 + LANES           # thread id in the workgroup of 256
    steps = run(prog, labels, w)
    if whole:                                   # everything the kernel may write: S, the selection records
        return np.frombuffer(mem[offs["S"]:total].tobytes(), np.uint32).copy(), steps
    return np.frombuffer(mem[offs["S"]:offs["S"] + sizes["S"]].tobytes(), np.uint32).copy(), steps


def main():
    if len(sys.argv) < 4 or sys.argv[1] != "pair":
        sys.exit(__doc__)
    a, b = sys.argv[2], sys.argv[3]
    prefix = sys.argv[4] if len(sys.argv) > 4 else "_ZN4wass6k_pairILi2ELi8ELi1EEE"
    bad = 0
    for (w1, h, dx, dy) in ((40, 36, 1, 1), (40, 36, -1, 1), (24, 30, 0, 1)):
        for chain in range(0, 4 * (w1 + h), 5):
            for end in (False, True):
                try:
                    sa, na = pair_case(a, prefix, w1, h, dx, dy, chain, 1234 + chain, with_endstate=end)
                    sb, nb = pair_case(b, prefix, w1, h, dx, dy, chain, 1234 + chain, with_endstate=end)
                except Unknown as e:
                    print("geometry", (w1, h, dx, dy), "chain", chain, "endstate", end, "->", e)
                    return 2
                diff = int((sa != sb).sum())
                bad += diff > 0
                print("geometry", (w1, h, dx, dy), "chain", chain, "endstate", end, "instructions", na, nb, "S dwords that differ:", diff, flush=True)
    print("listings disagree on", bad, "cases")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
