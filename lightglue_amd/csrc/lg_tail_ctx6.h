// EXPERIMENT BUILDS ONLY (lg_tail.hip with -DLG_EXPERIMENTS -DLG_TAIL_CTX_FP6=1, precision f16x3, 64-row tiles; never run on a GPU yet —
// DESIGN.md §7.1).  The ctx half of the fused tail's phase A as an f16 main product + fp6 (e2m3, MX block scales) cross terms
// whose hi-plane operands are DERIVED in registers from the f16 fragments.  Included twice by lg_tail.hip:
//   section 1: namespace-scope helpers;   section 2: the body of phase A, textually inside tail_kernel (it uses the kernel's
//   locals: acc, hreg, bf, load_half / store_half / load_b_A / read_af / mma_A, smem, t, tid, lane, w, lr, g).
// CPU-side checks of everything but the machine itself: tests/test_fp6_packing.py.
#if LG_TAIL_CTX6_SECTION == 1
// ---- fp6 (e2m3, MX block scaling) pieces of the ctx-half experiment.  Hardware facts they rely on (profiles/r02e_mfma_mx_probe.md,
// profiles/r02f_cvt_fp6_probe.md): lane (lr, g) of v_mfma_scale_f32_16x16x128_f8f6f4 supplies row lr, the 32 consecutive k of block g
// (slot i = bits [6i, 6i+6) of the 192-bit operand) and its own E8M0 scale; v_cvt_scalef32_pk32_fp6_f16 writes element i to slot i
// and returns fp6(x / scale); v_cvt_scalef32_2xpk16_fp6_f32(a0, a1) interleaves: slot 2i = a0[i], slot 2i+1 = a1[i].
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
// smallest E8M0 byte with amax / 2^(byte - 127) <= 7.5, the e2m3 maximum (as tools/ubench/ffn0_f16_fp6.hip, checked on the GPU)
__device__ __forceinline__ int e8m0_for(float amax) {
    if (amax == 0.f) return 127;
    const int e = (int)((__builtin_bit_cast(unsigned, amax * (16.f / 15.f)) >> 23) & 0xFF) - 2;
    return e < 1 ? 1 : e;
}
__device__ __forceinline__ v8i fp6_operand(const v6u& r) { return v8i{(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0}; }
// the fp6 copy of 32 f16 values held as four 16-byte fragments (element 8q + j = fragment q, element j), scale byte replicated in sb
__device__ __forceinline__ v8i fp6_from_f16x32(const u32x4& f0, const u32x4& f1, const u32x4& f2, const u32x4& f3, unsigned sb) {
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    const u32x16 all = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], f2[0], f2[1], f2[2], f2[3], f3[0], f3[1], f3[2], f3[3]};
    const float scale = __builtin_bit_cast(float, (sb & 0xFFu) << 23);            // 2^(byte - 127)
    return fp6_operand(__builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(v32h, all), scale));
}
// LDS record of one (row, 32-k block) of the ctx half: 24 B of fp6 lo values + lo scale dword + hi scale dword; 8 records per row,
// 32-byte slots XOR-swizzled with the row, the record's two 16-byte halves swapped on odd 8-row groups (conflict-free 16-lane reads)
__device__ __forceinline__ int ctx6_rec(int row, int blk) { return row * 256 + ((blk ^ (row & 7)) << 5); }
__device__ __forceinline__ int ctx6_half0(int row) { return ((row >> 3) & 1) << 4; }
#elif LG_TAIL_CTX6_SECTION == 2
        // ---- x half exactly as below (split-f16 x3, K-stages 0..3); ctx half: f16 main product + two fp6 cross terms per 128 k
        constexpr int HCX = NKC / 2;
        load_half(0);
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i) load_b_A(bf[i], i);
        store_half(0);
        __syncthreads();
        // ctx rows: thread = (row = tid >> 3, 32-k block = tid & 7), 32 consecutive floats (the MX block the fp6 conversion needs)
        const int crow = tid >> 3, cblk = tid & 7;
        {
            const float* src = a.CTX + (long long)(t.grow0 + crow) * 256 + 32 * cblk;
#pragma unroll
            for (int i = 0; i < 8; ++i) hreg[i >> 1][i & 1] = *reinterpret_cast<const f32x4*>(src + 4 * i);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 afx[2][MT][NPART];
        read_af(afx[0], 0);
#pragma unroll 1
        for (int c0 = 0; c0 < HCX; c0 += NBUF) {
#pragma unroll
            for (int i = 0; i < NBUF; ++i) {
                const int kc = c0 + i;
                load_b_A(bf[(i + NBUF - 1) % NBUF], kc + NBUF - 1 < HCX ? kc + NBUF - 1 : HCX - 1);
                read_af(afx[(i + 1) & 1], kc + 1 < HCX ? kc + 1 : kc);
                __builtin_amdgcn_sched_barrier(0);
                mma_A(afx[i & 1], bf[i]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- ctx tile -> LDS: f16 plane in K-stages 4..7 of the hi plane (standard layout), lo6 records in the lo plane's stages 4..7
        {
            float v[32], h[32];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const f32x4 q4 = hreg[i >> 1][i & 1]; v[4 * i] = q4[0]; v[4 * i + 1] = q4[1]; v[4 * i + 2] = q4[2]; v[4 * i + 3] = q4[3]; }
            float ah = 0.f, al = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) { h[i] = (float)(f16_t)v[i]; v[i] -= h[i]; ah = fmaxf(ah, fabsf(h[i])); al = fmaxf(al, fabsf(v[i])); }
#pragma unroll
            for (int q = 0; q < 4; ++q)   // k = 32 cblk + 8 q .. + 8: K-stage HS + (cblk >> 1), 16-byte slot (cblk & 1) * 4 + q
                *reinterpret_cast<u32x4*>(smem + (HS + (cblk >> 1)) * TILE + lds_off<128>(crow, (cblk & 1) * 4 + q)) =
                    u32x4{pack2_f16(h[8 * q], h[8 * q + 1]), pack2_f16(h[8 * q + 2], h[8 * q + 3]), pack2_f16(h[8 * q + 4], h[8 * q + 5]), pack2_f16(h[8 * q + 6], h[8 * q + 7])};
            const int sh = e8m0_for(ah), sl = e8m0_for(al);
            const float il = __builtin_bit_cast(float, (unsigned)(254 - sl) << 23);   // 2^-(sl - 127): exact pre-scaling, the conversion's own scale stays 1
            v16f e0, e1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { e0[i] = v[2 * i] * il; e1[i] = v[2 * i + 1] * il; }   // even / odd elements: the conversion interleaves its sources
            const v6u l6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(e0, e1, 1.0f);
            char* rec = smem + G_PLANE + HS * TILE + ctx6_rec(crow, cblk);
            const int h0 = ctx6_half0(crow);
            *reinterpret_cast<u32x4*>(rec + h0) = u32x4{l6[0], l6[1], l6[2], l6[3]};
            *reinterpret_cast<u32x4*>(rec + (h0 ^ 16)) = u32x4{l6[4], l6[5], (unsigned)sl * 0x01010101u, (unsigned)sh * 0x01010101u};
        }
        __syncthreads();
        // ---- ctx half.  k order of the f16 steps: step q of 128-k block c, lane (lr, g) <-> k = 128 c + 32 g + 8 q + j, so that a
        // lane's four fragments ARE the 32 consecutive k of its MX block g and the fp6 copy of a hi plane is ONE conversion of
        // registers the main product holds anyway (no bytes from L2 / LDS for it).  Per block c:
        //   stage A  for each of the wave's 4 n-tiles: 16 f16 MFMAs (4 k-steps x 4 row tiles) against the block's 16 resident
        //            activation fragments, then the n-tile's fp6 hi operand is derived from its four weight fragments;
        //   stage B  the activations' fp6 hi operands are derived, the lo6 records read, and the 2 x 16 fp6 MFMAs issued.
        // Weight fragments run on a 3-deep register ring (two n-tiles ~ 1k cycles ahead: an L2 round trip).
        const char* W16 = static_cast<const char*>(a.Wc16);   // [n-tile 32][c 2][q 4][lane 64][16 B] f16 hi, then [n-tile 32][c 2][lane 64] dwords: hi scale (E8M0, replicated)
        const char* W6 = static_cast<const char*>(a.Wc6);     // [n-tile 32][c 2][lane 64][32 B]: 24 B lo6, lo scale dword, pad
        constexpr long long W16S = 32LL * 2 * 4 * 64 * 16;    // byte offset of the hi-scale dwords
        auto load_wa = [&](u32x4 (&f)[4], unsigned& sh, int step) {   // step = 4 c + j (clamped by the caller)
            const long long base = (long long)((w + 8 * (step & 3)) * 2 + (step >> 2));
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = *reinterpret_cast<const u32x4*>(W16 + ((base * 4 + q) * 64 + lane) * 16);
            sh = *reinterpret_cast<const unsigned*>(W16 + W16S + (base * 64 + lane) * 4);
        };
        u32x4 wqf[3][4]; unsigned wqs[3];
        load_wa(wqf[0], wqs[0], 0); load_wa(wqf[1], wqs[1], 1);
        __builtin_amdgcn_sched_barrier(0);
        // (blocks and steps are spelled out with compile-time indices: with `step % 3` computed from loop variables the ring ended
        // up in scratch memory)
        auto block = [&](auto CC) {
            constexpr int c = decltype(CC)::value;
            u32x4 xa[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xa[mt][q] = *reinterpret_cast<const u32x4*>(smem + (HS + 2 * c + (g >> 1)) * TILE + lds_off<128>(mt * 16 + lr, (g & 1) * 4 + q));
            v8i wh6[4]; int swh[4]; u32x4 wr[4][2];
            auto stepA = [&](auto JJ) {
                constexpr int j = decltype(JJ)::value, step = 4 * c + j, cur = step % 3, nxt = (step + 2) % 3;
                load_wa(wqf[nxt], wqs[nxt], step + 2 < 8 ? step + 2 : 7);   // past the end: the last step again (no branch around a prefetch)
                {   // this n-tile's lo6 record, needed in stage B
                    const long long base = (long long)((w + 8 * j) * 2 + c);
                    wr[j][0] = *reinterpret_cast<const u32x4*>(W6 + (base * 64 + lane) * 32);
                    wr[j][1] = *reinterpret_cast<const u32x4*>(W6 + (base * 64 + lane) * 32 + 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mma_chunk<TagF16>(acc[mt][j], wqf[cur][q], xa[mt][q]);
                wh6[j] = fp6_from_f16x32(wqf[cur][0], wqf[cur][1], wqf[cur][2], wqf[cur][3], wqs[cur]);
                swh[j] = (int)wqs[cur];
                __builtin_amdgcn_sched_barrier(0);
            };
            stepA(std::integral_constant<int, 0>{}); stepA(std::integral_constant<int, 1>{});
            stepA(std::integral_constant<int, 2>{}); stepA(std::integral_constant<int, 3>{});
            // stage B
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = mt * 16 + lr;
                const char* rec = smem + G_PLANE + HS * TILE + ctx6_rec(row, 4 * c + g);
                const int h0 = ctx6_half0(row);
                const u32x4 r0 = *reinterpret_cast<const u32x4*>(rec + h0), r1 = *reinterpret_cast<const u32x4*>(rec + (h0 ^ 16));
                const v8i xl6 = v8i{(int)r0[0], (int)r0[1], (int)r0[2], (int)r0[3], (int)r1[0], (int)r1[1], 0, 0};
                const v8i xh6 = fp6_from_f16x32(xa[mt][0], xa[mt][1], xa[mt][2], xa[mt][3], r1[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v8i wl6 = v8i{(int)wr[j][0][0], (int)wr[j][0][1], (int)wr[j][0][2], (int)wr[j][0][3], (int)wr[j][1][0], (int)wr[j][1][1], 0, 0};
                    acc[mt][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wl6, xh6, acc[mt][j], 2, 2, 0, (int)wr[j][1][2], 0, (int)r1[3]);   // w_lo x_hi
                    acc[mt][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wh6[j], xl6, acc[mt][j], 2, 2, 0, swh[j], 0, (int)r1[2]);            // w_hi x_lo
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        block(std::integral_constant<int, 0>{});
        block(std::integral_constant<int, 1>{});
#endif
