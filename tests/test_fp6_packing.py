"""Host side of the ctx-half fp6 experiment (lg_engine.hip split_block_fp6, experiment builds only): the f16 / fp6 split of one
32-weight MX block against a numpy restatement of the same rule.  Needs the experiment library
(tools/build_variant.sh ctx6 -DLG_EXPERIMENTS -DLG_TAIL_CTX_FP6=1); skipped when it has not been built."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "lightglue_amd" / "liblightglue_amd_ctx6.so"


def e2m3_decode(code):
    s, e, m = (code >> 5) & 1, (code >> 3) & 3, code & 7
    v = m * 0.125 if e == 0 else (1.0 + m * 0.125) * 2.0 ** (e - 1)
    return -v if s else v


def tightest_scale(a):
    amax = np.abs(a).max()
    return 1.0 if amax == 0 else 2.0 ** np.ceil(np.log2(amax / 7.5))


def round_e2m3(x):
    """Round to the e2m3 grid (subnormal step 0.125 below 1.0), half to even like the hardware conversion, saturating at 7.5."""
    mag = np.abs(x)
    ex = np.maximum(np.floor(np.log2(np.where(mag > 0, mag, 1.0))), 0)
    step = 2.0 ** (ex - 3)
    return np.clip(np.round(x / step) * step, -7.5, 7.5)


@pytest.mark.skipif(not LIB.exists(), reason="experiment library not built")
def test_split_block_matches_the_numpy_restatement():
    lib = C.CDLL(str(LIB))
    f = lib.lg_debug_split_block_fp6
    f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_uint16), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
    rng = np.random.default_rng(0)
    for trial in range(300):
        v = (rng.standard_normal(32) * 10.0 ** rng.uniform(-4, 1)).astype(np.float32)
        if trial == 0:
            v[:] = 0
        if trial == 1:
            v[:] = 0
            v[5] = 0.3   # one non-zero element
        h16 = (C.c_uint16 * 32)(); lo6 = (C.c_uint32 * 6)(); sh = C.c_int32(); sl = C.c_int32()
        assert f(v.ctypes.data_as(C.POINTER(C.c_float)), h16, C.byref(sh), lo6, C.byref(sl)) == 0
        hi = np.frombuffer(bytes(h16), np.float16).astype(np.float32)
        np.testing.assert_array_equal(hi, v.astype(np.float16).astype(np.float32))
        lo_true = (v - hi).astype(np.float64)                      # exact in fp32
        s_lo, s_hi = 2.0 ** (sl.value - 127), 2.0 ** (sh.value - 127)
        bits = int.from_bytes(bytes(lo6), "little")
        lo = np.array([e2m3_decode((bits >> (6 * i)) & 63) for i in range(32)], np.float64) * s_lo
        # scales: never clip, and at most one binade above the tightest power of two (the kernel-side rule is one binade loose
        # only exactly at amax = 7.5 * 2^k)
        for a, s in ((lo_true, s_lo), (hi, s_hi)):
            if np.abs(a).max() > 0 and s > 2.0 ** -126:
                assert np.abs(a).max() / s <= 7.5
                assert s in (tightest_scale(a), 2 * tightest_scale(a))
        # codes: exactly the round-to-nearest-even e2m3 value of lo / scale
        np.testing.assert_array_equal(lo, round_e2m3(lo_true / s_lo) * s_lo)


# ---------------------------------------------------------------------------------------------------------------------------
# Lane-level walk through the ctx half of lg_tail.hip's LG_TAIL_CTX_FP6 path, on the CPU: the REAL host packer's buffers, the
# kernel's LDS store / read address formulas and operand layouts restated in numpy, MFMA semantics as confirmed on the GPU
# (profiles/r02e_mfma_mx_probe.md, r02f_cvt_fp6_probe.md).  It cannot replace a GPU run; it pins the one class of bug a GPU run
# would otherwise spend minutes on — an index that disagrees between the packer, the LDS writer and the fragment reader.
TILE, G_PLANE, HS = 4 * 16 * 128, 8 * 4 * 16 * 128, 4


def lds_off128(row, slot):
    return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4)


def ctx6_rec(row, blk):
    return row * 256 + ((blk ^ (row & 7)) << 5)


def e8m0_for(amax):
    if amax == 0:
        return 127
    e = ((np.float32(np.float32(amax) * np.float32(16.0 / 15.0)).view(np.uint32) >> 23) & 0xFF) - 2
    return max(int(e), 1)


def pack_fp6(vals):
    """32 values on the e2m3 grid -> 192 bits, slot i at bits [6i, 6i+6)."""
    bits = 0
    for i, v in enumerate(vals):
        a = abs(float(v))
        if a < 1.0:
            m = int(round(a * 8)); code = 8 if m == 8 else m
        else:
            e = 0 if a < 2 else (1 if a < 4 else 2)
            code = ((e + 1) << 3) | (int(round(a / 2.0 ** (e - 3))) - 8)
        if np.signbit(v):
            code |= 32
        bits |= code << (6 * i)
    return bits


def unpack_fp6(bits):
    return np.array([e2m3_decode((bits >> (6 * i)) & 63) for i in range(32)], np.float64)


@pytest.mark.skipif(not LIB.exists(), reason="experiment library not built")
def test_ctx_half_data_flow_lane_by_lane():
    lib = C.CDLL(str(LIB))
    pk = lib.lg_debug_pack_ctx6
    pk.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    n16, n6 = C.c_int64(), C.c_int64()
    pk(None, None, None, C.byref(n16), C.byref(n6))
    rng = np.random.default_rng(3)
    cat = (rng.standard_normal((512, 512)) * 0.05).astype(np.float32).astype(np.float64)       # fp32-representable, as the engine folds and stores it
    ctx = (rng.standard_normal((64, 256)) * rng.choice([0.3, 2.0], size=(64, 256))).astype(np.float32)
    w16 = np.zeros(n16.value, np.uint8); w6 = np.zeros(n6.value, np.uint8)
    assert pk(cat.ctypes.data, w16.ctypes.data, w6.ctypes.data, None, None) == 0
    W16S = 32 * 2 * 4 * 64 * 16

    # ---- the workgroup's LDS after the ctx store (thread = (row = tid >> 3, block = tid & 7))
    lds = np.zeros(2 * G_PLANE, np.uint8)
    written = np.zeros(2 * G_PLANE, bool)

    def put(off, raw):
        assert not written[off:off + len(raw)].any(), "two threads write the same LDS bytes"
        lds[off:off + len(raw)] = np.frombuffer(raw, np.uint8); written[off:off + len(raw)] = True
    for tid in range(512):
        crow, cblk = tid >> 3, tid & 7
        v = ctx[crow, 32 * cblk:32 * cblk + 32]
        h = v.astype(np.float16)
        lo = (v - h.astype(np.float32)).astype(np.float64)
        sh, sl = e8m0_for(np.abs(h.astype(np.float32)).max()), e8m0_for(np.abs(lo).max())
        for q in range(4):
            put((HS + (cblk >> 1)) * TILE + lds_off128(crow, (cblk & 1) * 4 + q), h[8 * q:8 * q + 8].tobytes())
        l6 = pack_fp6(round_e2m3(lo * 2.0 ** (127 - sl)))
        rec = l6.to_bytes(24, "little") + (sl * 0x01010101).to_bytes(4, "little") + (sh * 0x01010101).to_bytes(4, "little")
        base = G_PLANE + HS * TILE + ctx6_rec(crow, cblk)
        h0 = ((crow >> 3) & 1) << 4
        put(base + h0, rec[:16]); put(base + (h0 ^ 16), rec[16:])

    # ---- every wave / n-tile / block / row tile as the kernel's fragment reads see them
    H = np.zeros((64, 512))
    lanes = np.arange(64); lr = lanes & 15; g = lanes >> 4
    for w in range(8):
        for j in range(4):
            nt = w + 8 * j
            for c in range(2):
                base = nt * 2 + c
                # weights, per lane: 4 f16 fragments, hi scale dword, lo6 record
                wf = np.zeros((64, 32), np.float64); wh6 = np.zeros((64, 32)); wl6 = np.zeros((64, 32)); swl = np.zeros(64); swh = np.zeros(64)
                for ln in range(64):
                    for q in range(4):
                        o = ((base * 4 + q) * 64 + ln) * 16
                        wf[ln, 8 * q:8 * q + 8] = w16[o:o + 16].view(np.float16).astype(np.float64)
                    shb = int(w16[W16S + (base * 64 + ln) * 4])
                    assert bytes(w16[W16S + (base * 64 + ln) * 4:W16S + (base * 64 + ln) * 4 + 4]) == bytes([shb] * 4)
                    rec = bytes(w6[(base * 64 + ln) * 32:(base * 64 + ln) * 32 + 32])
                    slb = rec[24]
                    wl6[ln] = unpack_fp6(int.from_bytes(rec[:24], "little")); swl[ln] = 2.0 ** (slb - 127)
                    wh6[ln] = round_e2m3(wf[ln] / 2.0 ** (shb - 127)); swh[ln] = 2.0 ** (shb - 127)          # the conversion the kernel derives it with
                    # the fragments must be exactly this lane's 32 consecutive k of row nt*16+lr, MX block g
                    k0 = 256 + 128 * c + 32 * (ln >> 4)
                    np.testing.assert_array_equal(wf[ln], cat[nt * 16 + (ln & 15), k0:k0 + 32].astype(np.float16).astype(np.float64))
                for mt in range(4):
                    xf = np.zeros((64, 32)); xl6 = np.zeros((64, 32)); xh6 = np.zeros((64, 32)); sxl = np.zeros(64); sxh = np.zeros(64)
                    for ln in range(64):
                        row = mt * 16 + (ln & 15); gg = ln >> 4
                        for q in range(4):
                            o = (HS + 2 * c + (gg >> 1)) * TILE + lds_off128(row, (gg & 1) * 4 + q)
                            assert written[o:o + 16].all()
                            xf[ln, 8 * q:8 * q + 8] = lds[o:o + 16].view(np.float16).astype(np.float64)
                        rb = G_PLANE + HS * TILE + ctx6_rec(row, 4 * c + gg)
                        h0 = ((row >> 3) & 1) << 4
                        r0 = bytes(lds[rb + h0:rb + h0 + 16]); r1 = bytes(lds[rb + (h0 ^ 16):rb + (h0 ^ 16) + 16])
                        xl6[ln] = unpack_fp6(int.from_bytes(r0 + r1[:8], "little"))
                        sxl[ln] = 2.0 ** (r1[8] - 127); sxh[ln] = 2.0 ** (r1[12] - 127)
                        xh6[ln] = round_e2m3(xf[ln] / sxh[ln])
                        k0 = 128 * c + 32 * gg
                        np.testing.assert_array_equal(xf[ln], ctx[row, k0:k0 + 32].astype(np.float16).astype(np.float64))
                    # D[a][b] = sum over g of A(lane a, g) . B(lane b, g): 16 x 16 tile, weights = A operand
                    D = np.zeros((16, 16))
                    for gg in range(4):
                        A = wf[gg * 16:(gg + 1) * 16]; B = xf[gg * 16:(gg + 1) * 16]
                        D += A @ B.T
                        D += (wl6[gg * 16:(gg + 1) * 16] * swl[gg * 16:(gg + 1) * 16, None]) @ (xh6[gg * 16:(gg + 1) * 16] * sxh[gg * 16:(gg + 1) * 16, None]).T
                        D += (wh6[gg * 16:(gg + 1) * 16] * swh[gg * 16:(gg + 1) * 16, None]) @ (xl6[gg * 16:(gg + 1) * 16] * sxl[gg * 16:(gg + 1) * 16, None]).T
                    H[mt * 16:(mt + 1) * 16, nt * 16:(nt + 1) * 16] += D.T
    ref = ctx.astype(np.float64) @ cat[:, 256:].T
    mag = np.abs(ctx.astype(np.float64)) @ np.abs(cat[:, 256:]).T
    err = np.abs(H - ref) / mag
    assert err.max() < 2e-5, err.max()          # a dropped cross term shows ~2e-4, a wrong index O(1); the scheme itself ~5e-6
