"""A small lane-level interpreter for the gfx950 instructions the generated FLL assembly uses (csrc/fll*_asm.inc), so that the
CPU suite can execute the generated TEXT -- registers, packed-FP32 operand selection, DPP row shifts, LDS traffic, the scalar
loop control -- and compare what it computes with the oracle, without a GPU.  Test infrastructure only.

Arithmetic is IEEE binary32 with a correctly rounded fma (C fmaf through a helper compiled on the fly), i.e. what the
hardware's v_fma / v_pk_fma / v_fmac do; v_rndne is round-half-even; v_med3 the median; DPP shifts stay inside 16-lane rows
and either keep the destination (bound_ctrl off) or write zero (bound_ctrl:1) where the source lane does not exist.
Only what the blocks contain is implemented; anything else raises."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

_HELPER = None


def _helper():
    global _HELPER
    if _HELPER is None:
        d = tempfile.mkdtemp(prefix="gcn_sim_")
        src = os.path.join(d, "fma32.c")
        with open(src, "w") as f:
            f.write("#include <math.h>\nvoid fma32(const float* a, const float* b, const float* c, float* o, int n) {\n"
                    "    for (int i = 0; i < n; i++) o[i] = fmaf(a[i], b[i], c[i]);\n}\n")
        lib = os.path.join(d, "libfma32.so")
        subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", lib, src, "-lm"], check=True)
        L = C.CDLL(lib)
        L.fma32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.fma32.restype = None
        _HELPER = L
    return _HELPER


def fma(a, b, c):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    c = np.ascontiguousarray(c, np.float32)
    o = np.empty_like(a)
    _helper().fma32(a.ctypes.data, b.ctypes.data, c.ctypes.data, o.ctypes.data, a.size)
    return o


def f2u(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def u2f(x):
    return np.ascontiguousarray(x, np.uint32).view(np.float32)


def parse_block(path, macro):
    """The instruction lines and labels of `#define <macro>_ASM` and the four 64-bit Horner constants."""
    src = open(path).read()
    body = src[src.index("#define %s_ASM" % macro):]
    body = body[:body.index("\n#define %s_CLOBBERS" % macro)]
    lines = [ln.strip() for ln in re.findall(r'"([^"]*)\\n"', body)]
    consts = {k: int(v, 16) for k, v in re.findall(r"#define %s_(K[1-4]) (0x[0-9a-f]+)ull" % macro, src)}
    return lines, consts


class Sim:
    LANES = 64

    def __init__(self, lines, vec_in, sca_in, lds, on_barrier=None):
        """vec_in: name -> uint32[64] (the "v" operands); sca_in: name -> int (32-bit) or (lo, hi) for 64-bit "s" operands;
        lds: np.uint8 array; on_barrier(sim, count) is called at every s_barrier."""
        self.lines = lines
        self.labels = {ln[:-1]: i for i, ln in enumerate(lines) if ln.endswith(":")}
        self.V = np.zeros((256, self.LANES), np.uint32)
        self.vec = {k: np.array(v, np.uint32).copy() for k, v in vec_in.items()}
        self.sca = dict(sca_in)
        self.lds = lds
        self.vcc = np.zeros(self.LANES, bool)
        self.scc = False
        self.on_barrier = on_barrier
        self.barriers = 0
        self.executed = 0

    # ---- operands -------------------------------------------------------------------------------------------------
    def _src32(self, tok, as_float=True):
        tok = tok.strip()
        neg = tok.startswith("-") and not re.match(r"^-\d", tok)
        if neg:
            tok = tok[1:]
        ab = tok.startswith("|")
        if ab:
            tok = tok.strip("|")
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            v = self.V[int(m.group(1))].copy()
        elif tok.startswith("%["):
            name = tok[2:-1]
            if name in self.vec:
                v = self.vec[name].copy()
            else:
                s = self.sca[name]
                v = np.full(self.LANES, (s[0] if isinstance(s, tuple) else s) & 0xffffffff, np.uint32)
        elif tok.startswith("0x"):
            v = np.full(self.LANES, int(tok, 16), np.uint32)
        elif re.fullmatch(r"-?\d+\.\d+", tok):
            v = np.full(self.LANES, np.float32(float(tok)).view(np.uint32), np.uint32)
        elif re.fullmatch(r"-?\d+", tok):
            v = np.full(self.LANES, (np.float32(int(tok)).view(np.uint32) if as_float else np.uint32(int(tok) & 0xffffffff)), np.uint32)
        else:
            raise ValueError("operand " + tok)
        if ab:
            v &= np.uint32(0x7fffffff)
        if neg:
            v ^= np.uint32(0x80000000)
        return v

    def _src64(self, tok):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            a = int(m.group(1))
            return self.V[a].copy(), self.V[a + 1].copy()
        if tok.startswith("%["):
            s = self.sca[tok[2:-1]]
            lo, hi = s if isinstance(s, tuple) else (s & 0xffffffff, (s >> 32) & 0xffffffff)
            return np.full(self.LANES, lo, np.uint32), np.full(self.LANES, hi, np.uint32)
        v = self._src32(tok)                     # inline constant: both halves
        return v, v.copy()

    @staticmethod
    def _mods(rest):
        out = {}
        for name in ("op_sel_hi", "op_sel", "neg_lo", "neg_hi"):
            m = re.search(r"\b%s:\[([0-9,]+)\]" % name, rest)
            if m:
                out[name] = [int(x) for x in m.group(1).split(",")]
        return out

    @staticmethod
    def _split(rest):
        head = re.split(r"\s(?:op_sel|op_sel_hi|neg_lo|neg_hi|row_shr|row_shl|row_mask|bank_mask|bound_ctrl|offset):", " " + rest + " ")[0]
        return [o.strip() for o in re.split(r",(?![^\[]*\])", head.strip())]

    def _dst(self, tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok.strip())
        if m:
            return int(m.group(1))
        return int(re.fullmatch(r"v(\d+)", tok.strip()).group(1))

    # ---- LDS --------------------------------------------------------------------------------------------------------
    def _lds_read(self, addr, nbytes):
        idx = addr.astype(np.int64)[:, None] + np.arange(nbytes)[None, :]
        return self.lds[idx].reshape(self.LANES, nbytes // 4, 4).copy().view(np.uint32).reshape(self.LANES, nbytes // 4)

    def _lds_write(self, addr, words):
        b = np.ascontiguousarray(words, np.uint32).view(np.uint8).reshape(self.LANES, -1)
        for lane in range(self.LANES):             # lane order: a later lane wins, like the hardware for colliding addresses
            a = int(addr[lane])
            self.lds[a:a + b.shape[1]] = b[lane]

    # ---- execution ----------------------------------------------------------------------------------------------------
    def run(self, max_steps=5000000):
        pc = 0
        while pc < len(self.lines):
            ln = self.lines[pc]
            pc += 1
            if ln.endswith(":"):
                continue
            self.executed += 1
            assert self.executed < max_steps, "runaway"
            op, _, rest = ln.partition(" ")
            jump = self._exec(op, rest)
            if jump is not None:
                pc = self.labels[jump]
        return self

    def _exec(self, op, rest):
        V = self.V
        if op in ("s_nop", "s_waitcnt"):
            return None
        if op == "s_barrier":
            self.barriers += 1
            if self.on_barrier:
                self.on_barrier(self, self.barriers)
            return None
        o = self._split(rest)
        if op.startswith("s_"):
            def sval(t):
                t = t.strip()
                if t.startswith("%["):
                    return int(self.sca[t[2:-1]]) & 0xffffffff
                return int(t, 0) & 0xffffffff
            if op == "s_mov_b32":
                self.sca[o[0][2:-1]] = sval(o[1])
            elif op == "s_and_b32":
                self.sca[o[0][2:-1]] = sval(o[1]) & sval(o[2])
            elif op == "s_lshl_b32":
                self.sca[o[0][2:-1]] = (sval(o[1]) << sval(o[2])) & 0xffffffff
            elif op == "s_add_u32":
                self.sca[o[0][2:-1]] = (sval(o[1]) + sval(o[2])) & 0xffffffff
            elif op == "s_sub_u32":
                self.sca[o[0][2:-1]] = (sval(o[1]) - sval(o[2])) & 0xffffffff
            elif op == "s_cmp_lg_u32":
                self.scc = sval(o[0]) != sval(o[1])
            elif op == "s_cbranch_scc1":
                return o[0] if self.scc else None
            else:
                raise ValueError(op)
            return None
        if op.startswith("ds_read_b"):
            n = int(op[len("ds_read_b"):]) // 8
            m = re.search(r"offset:(\d+)", rest)
            addr = self._src32(o[1], as_float=False) + np.uint32(int(m.group(1)) if m else 0)
            w = self._lds_read(addr, n)
            d = self._dst(o[0])
            for k in range(n // 4):
                V[d + k] = w[:, k]
            return None
        if op == "ds_write_b64":
            m = re.search(r"offset:(\d+)", rest)
            addr = self._src32(o[0], as_float=False) + np.uint32(int(m.group(1)) if m else 0)
            lo, hi = self._src64(o[1])
            self._lds_write(addr, np.stack([lo, hi], axis=1))
            return None
        if op == "v_mov_b32_dpp":
            d, src = self._dst(o[0]), self._src32(o[1])
            shr = re.search(r"row_shr:(\d+)", rest)
            shl = re.search(r"row_shl:(\d+)", rest)
            zero = "bound_ctrl:1" in rest
            assert "row_mask:0xf" in rest and "bank_mask:0xf" in rest
            lane = np.arange(self.LANES)
            inrow = lane & 15
            if shr:
                h = int(shr.group(1))
                ok, from_ = inrow >= h, lane - h
            else:
                h = int(shl.group(1))
                ok, from_ = inrow < 16 - h, lane + h
            res = np.where(ok, src[np.clip(from_, 0, self.LANES - 1)], np.uint32(0) if zero else V[d])
            V[d] = res
            return None
        if op.startswith("v_pk_"):
            mods = self._mods(rest)
            nsrc = 3 if op == "v_pk_fma_f32" else 2
            osel = mods.get("op_sel", [0] * nsrc) + [0] * 3
            ohi = mods.get("op_sel_hi", [1] * nsrc) + [1] * 3
            nlo = mods.get("neg_lo", [0] * nsrc) + [0] * 3
            nhi = mods.get("neg_hi", [0] * nsrc) + [0] * 3
            srcs = [self._src64(t) for t in o[1:1 + nsrc]]
            lo_ops = [u2f(srcs[i][osel[i]] ^ np.uint32(0x80000000 if nlo[i] else 0)) for i in range(nsrc)]
            hi_ops = [u2f(srcs[i][ohi[i]] ^ np.uint32(0x80000000 if nhi[i] else 0)) for i in range(nsrc)]
            if op == "v_pk_fma_f32":
                rl, rh = fma(*lo_ops), fma(*hi_ops)
            elif op == "v_pk_mul_f32":
                rl, rh = lo_ops[0] * lo_ops[1], hi_ops[0] * hi_ops[1]
            elif op == "v_pk_add_f32":
                rl, rh = lo_ops[0] + lo_ops[1], hi_ops[0] + hi_ops[1]
            else:
                raise ValueError(op)
            d = self._dst(o[0])
            V[d], V[d + 1] = f2u(rl), f2u(rh)
            return None
        if op == "v_mov_b64":
            d = self._dst(o[0])
            assert o[1] == "0"
            V[d] = 0
            V[d + 1] = 0
            return None
        if op == "v_mov_b32":
            if o[0].startswith("%["):                      # result operands of the block
                self.vec[o[0][2:-1]] = self._src32(o[1])
            else:
                V[self._dst(o[0])] = self._src32(o[1])
            return None
        if op == "v_add_u32":
            V[self._dst(o[0])] = self._src32(o[1], as_float=False) + self._src32(o[2], as_float=False)
            return None
        if op == "v_sub_u32":
            V[self._dst(o[0])] = (self._src32(o[1], as_float=False) - self._src32(o[2], as_float=False)) & np.uint32(0xffffffff)
            return None
        if op == "v_xor_b32":
            V[self._dst(o[0])] = self._src32(o[1], as_float=False) ^ self._src32(o[2], as_float=False)
            return None
        if op == "v_bfi_b32":
            m, a, b = (self._src32(t, as_float=False) for t in o[1:4])
            V[self._dst(o[0])] = (m & a) | (~m & b)
            return None
        if op == "v_cmp_gt_f32":
            assert o[0] == "vcc"
            self.vcc = u2f(self._src32(o[1])) > u2f(self._src32(o[2]))
            return None
        if op == "v_cndmask_b32":
            assert o[3] == "vcc"
            V[self._dst(o[0])] = np.where(self.vcc, self._src32(o[2]), self._src32(o[1]))
            return None
        d = self._dst(o[0])
        s = [u2f(self._src32(t)) for t in o[1:]]
        with np.errstate(all="ignore"):
            if op == "v_mul_f32":
                r = s[0] * s[1]
            elif op == "v_add_f32":
                r = s[0] + s[1]
            elif op == "v_sub_f32":
                r = s[0] - s[1]
            elif op == "v_max_f32":
                r = np.maximum(s[0], s[1])
            elif op == "v_min_f32":
                r = np.minimum(s[0], s[1])
            elif op == "v_rndne_f32":
                r = np.rint(s[0])
            elif op == "v_fma_f32":
                r = fma(s[0], s[1], s[2])
            elif op == "v_fmac_f32":
                r = fma(s[0], s[1], u2f(V[d]))
            elif op == "v_fmamk_f32":                      # D = S0 * K + S1
                r = fma(s[0], s[1], s[2])
            elif op == "v_med3_f32":
                r = np.maximum(np.minimum(s[0], s[1]), np.minimum(np.maximum(s[0], s[1]), s[2]))
            else:
                raise ValueError(op)
        V[d] = f2u(r.astype(np.float32))
        return None
