// Block-parallel greedy resolution, four chains per wave (gfx950).
//
// k_resolve_blocks (swsem_kernels.hip) gives every chain a whole wave: the automaton's values are wave-uniform and
// live in scalar registers, and the 64 lanes only serve the two wide loads of a step. Measured on configs[2]
// (profiles/r02_resolve_phases.json) a chain spends 43 % of its cycles waiting for a window's table gathers, 14 %
// for a visit's reference bytes and 40 % issuing the automaton's scalar instructions — it is bound by the latency
// of its own dependent chain, with the memory side at 60 % of its random-sector rate and 63 of 64 lanes idle.
// Here a wave runs FOUR chains, 16 lanes each: the automaton's values are uniform inside a 16-lane group and held in
// vector registers, the four groups execute one instruction stream (the same refill / visit / resolve step for
// whichever of them needs it), and every wait is shared by four chains. A scan window is 16 positions (one per
// lane: the reference is sampled every 16th position, so the next hit is at most 16 clean positions away), a visit
// loads 16 bytes per lane and side (64 to the left of the candidate, 192 from it on). The four chains of a wave
// scan the same offsets of four different contigs (run_batch): on a collection their gathers largely fall into the
// same sectors and leave the wave as one request.
//
// Semantics are those of run_chain_lazy / visit / process_hit — SlidingWindowSparseEMMatcher.cpp:200-315 — and the
// block protocol (warm-up, boundary snapshot, BlockRec) is k_resolve_blocks'; k_stitch_pre / k_stitch / k_gather
// consume the records unchanged, so results equal the sequential loop by construction whatever this kernel speculates.
#include "swsem_device.h"

namespace swk {

constexpr int GL = 16;                 // lanes of a chain
constexpr int GC = WAVE / GL;          // chains per wave
#ifndef SWSEM_WIN4
#define SWSEM_WIN4 16
#endif
constexpr int WN4 = SWSEM_WIN4;          // positions looked up per window (<= GL)
constexpr int RING4 = 16;              // newest rows of a chain's stack mirrored in LDS
constexpr int LEFT4 = 64;              // bytes left of the candidate covered by a visit's first load
constexpr int LL4 = LEFT4 / 16;        // lanes holding them
#ifndef SWSEM_VISIT_LANES
#define SWSEM_VISIT_LANES 16
#endif
constexpr int NL4 = SWSEM_VISIT_LANES; // lanes whose 16 reference bytes a visit's first load fetches (the rest is fetched when the run gets there)
#ifndef SWSEM_RESOLVE4_WAVES
#define SWSEM_RESOLVE4_WAVES 6
#endif
constexpr int RESOLVE4_WAVES_PER_SIMD = SWSEM_RESOLVE4_WAVES;   // the register budget the kernel is compiled for (80 vector registers at 6)
constexpr int K_MAX4 = 40;             // window bytes GL + K - 1 (+3 of misalignment) must fit the group's 16 dwords

__device__ __forceinline__ uint32_t gballot(bool p, uint32_t gbase) { return (uint32_t) (__ballot(p) >> gbase) & 0xFFFFu; }
__device__ __forceinline__ uint32_t gread(uint32_t x, uint32_t srcLane) {
    return (uint32_t) __builtin_amdgcn_ds_bpermute((int) (srcLane << 2), (int) x);
}
__device__ __forceinline__ bool nz128(const uint4 &x) { return (x.x | x.y | x.z | x.w) != 0; }
__device__ __forceinline__ uint4 xor128(const uint4 &a, const uint4 &b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
// equal bytes of a 16-byte chunk counted from its low end / from its high end (x = a ^ b)
__device__ __forceinline__ uint32_t eq_low(const uint4 &x) {
    if (x.x) return (uint32_t) __builtin_ctz(x.x) >> 3;
    if (x.y) return 4u + ((uint32_t) __builtin_ctz(x.y) >> 3);
    if (x.z) return 8u + ((uint32_t) __builtin_ctz(x.z) >> 3);
    if (x.w) return 12u + ((uint32_t) __builtin_ctz(x.w) >> 3);
    return 16u;
}
__device__ __forceinline__ uint32_t eq_high(const uint4 &x) {
    if (x.w) return (uint32_t) __builtin_clz(x.w) >> 3;
    if (x.z) return 4u + ((uint32_t) __builtin_clz(x.z) >> 3);
    if (x.y) return 8u + ((uint32_t) __builtin_clz(x.y) >> 3);
    if (x.x) return 12u + ((uint32_t) __builtin_clz(x.x) >> 3);
    return 16u;
}

// group-cooperative continuation of a run, 256 bytes per step (wave_lcp_fwd / _bwd with 16 lanes x 16 bytes):
// equal bytes of a[n..limit) vs b[n..limit) given that the first n are equal
__device__ uint32_t group_lcp_fwd(const uint8_t *a, const uint8_t *b, uint32_t n, uint32_t limit, uint32_t gl, uint32_t gbase) {
    while (n < limit) {
        const uint32_t off = n + 16u * gl;
        uint32_t eq = 16;
        bool stop = false;
        if (off + 16 <= limit) {
            const uint4 x = xor128(ld_u128(a + off), ld_u128(b + off));
            if (nz128(x)) { eq = eq_low(x); stop = true; }
        } else {
            eq = 0; stop = true;
            if (off < limit) {
                const uint32_t avail = limit - off;
                while (eq < avail && a[off + eq] == b[off + eq]) eq++;
            }
        }
        const uint32_t bal = gballot(stop, gbase);
        if (bal) {
            const uint32_t l = (uint32_t) __builtin_ctz(bal);
            return n + 16u * l + gread(eq, gbase + l);
        }
        n += 16u * GL;
    }
    return limit;
}
// the same to the left: equal bytes of a[-1-k] vs b[-1-k], k in [n, limit)
__device__ uint32_t group_lcp_bwd(const uint8_t *a, const uint8_t *b, uint32_t n, uint32_t limit, uint32_t gl, uint32_t gbase) {
    while (n < limit) {
        const uint32_t off = n + 16u * gl;
        uint32_t eq = 16;
        bool stop = false;
        if (off + 16 <= limit) {
            const uint4 x = xor128(ld_u128(a - off - 16), ld_u128(b - off - 16));
            if (nz128(x)) { eq = eq_high(x); stop = true; }
        } else {
            eq = 0; stop = true;
            if (off < limit) {
                const uint32_t avail = limit - off;
                while (eq < avail && a[-(int64_t) (off + eq) - 1] == b[-(int64_t) (off + eq) - 1]) eq++;
            }
        }
        const uint32_t bal = gballot(stop, gbase);
        if (bal) {
            const uint32_t l = (uint32_t) __builtin_ctz(bal);
            return n + 16u * l + gread(eq, gbase + l);
        }
        n += 16u * GL;
    }
    return limit;
}

// order4[slot * GC + k] = resolve block (cg.rb0 + b) run by chain k of wave `slot`, 0xFFFFFFFF = none
template <bool LAPS>
__global__ void __launch_bounds__(WAVE, SWSEM_RESOLVE4_WAVES) k_resolve_blocks4(RefView v, const uint8_t *__restrict__ qbuf,
                                                          const Contig *__restrict__ contigs,
                                                          const uint32_t *__restrict__ rbContig,
                                                          const uint32_t *__restrict__ order4,
                                                          Row *__restrict__ regions, uint32_t cap, uint32_t rb,
                                                          BlockRec *__restrict__ recs, uint32_t overlap) {
    __shared__ uint4 ring[GC][RING4];              // {posDest, len, posSrc lo, posSrc hi} of the newest rows
    __shared__ __attribute__((aligned(16))) uint8_t qcache[GC][16 * GL + 16];   // query bytes [qb0, qb0 + 256) of every chain (+ slack: a window's last dword)
    const uint32_t lane = threadIdx.x, grp = lane >> 4, gl = lane & 15u, gbase = lane & 48u;
    const uint32_t g = order4[blockIdx.x * GC + grp];
    const int32_t K = v.K;
    int32_t phase = 2;                             // 0 warm-up on the predecessor's tail, 1 own tiles, 2 done / no block
    const uint8_t *q = qbuf;
    uint32_t n = 0;
    uint64_t lock = UINT64_MAX;
    int32_t w0 = 0, w1 = 0, p0 = 0, p1 = 0, scan = 0;
    Row *st = regions;
    if (g != 0xFFFFFFFFu) {
        const Contig *cg = contigs + rbContig[g];
        q = qbuf + cg->qoff;
        n = (uint32_t) cg->n;
        lock = cg->lock;
        const uint32_t b = g - cg->rb0;
        const int32_t npos = n >= (uint32_t) K ? (int32_t) (n - (uint32_t) K + 1u) : 0;
        w0 = (int32_t) (b * rb * RBU);
        w1 = w0 + (int32_t) (rb * RBU) < npos ? w0 + (int32_t) (rb * RBU) : npos;
        st = regions + (uint64_t) g * cap;
        phase = 0;
        p0 = b && w0 > (int32_t) overlap ? w0 - (int32_t) overlap : 0;   // block 0 starts from the true (empty) state: its warm-up is empty
        p1 = w0;
        scan = p0;
    }
    if (__ballot(phase != 2) == 0) return;
    const uint64_t tstart = __builtin_amdgcn_s_memtime();
    int32_t sp = 0, ringLow = 0;
    int32_t minTouched = 0x7fffffff, minKeep = 0x7fffffff;   // (minTouched stays at its start value exactly while no hit has been processed)
    int32_t wb = -0x40000000;                      // scan window [wb, wb + GL)
    uint32_t went = 0, m16 = 0;                    // this lane's table value in the window / the group's candidate lanes
    int32_t qb0 = 0, qlo = 0, qhi = 0;             // query cache: first byte position of the buffer, valid positions [qlo, qhi)
#ifndef SWSEM_SYNC_SPAN
#define SWSEM_SYNC_SPAN 0
#endif
    // The four chains of a wave scan the same offsets of four contigs; kept within SWSEM_SYNC_SPAN positions of each
    // other (a chain that has run ahead waits at the next multiple) their lookups keep falling into the same sectors.
    int32_t syncRel = SWSEM_SYNC_SPAN;             // wave-uniform: no window is opened at or beyond this offset from the chain's first position
    const int32_t start0 = scan;

    while (true) {
        // ---- leaving a range: the block boundary (snapshot), then the block's end (record)
#pragma unroll
        for (int rep = 0; rep < 2; rep++) {
            if (phase < 2) {
                const int32_t s = scan > p0 ? scan : p0;
                if (s < p1) scan = s;
                else {
                    __builtin_amdgcn_s_waitcnt(0);                 // the chain's own rows are read back below
                    BlockRec *rec = recs + g;
                    Match *top = phase == 0 ? rec->bTop : rec->fTop;
                    if (gl < 3u * SNAP) {                          // newest rows, newest first, one u64 per lane
                        const int32_t j = (int32_t) (gl / 3u), f = (int32_t) (gl % 3u);
                        ((uint64_t *) top)[gl] = j < sp ? ((const uint64_t *) (st + (sp - 1 - j)))[f] : 0ull;
                    }
                    if (phase == 0) {
                        if (gl == 0) { rec->scanB = scan > w0 ? scan : w0; rec->spB = sp; }
                        minTouched = 0x7fffffff; minKeep = sp;
                        phase = 1; p0 = w0; p1 = w1;
                        wb = -0x40000000;                          // positions from w0 on were not looked up
                    } else {
                        if (gl == 0) {
                            const int32_t spB = rec->spB;                    // (written at the boundary by this lane)
                            rec->scanF = scan; rec->spF = sp;
                            rec->minTouched = minTouched;
                            rec->minKeep = minKeep < spB ? minKeep : spB;
                            rec->cycles = __builtin_amdgcn_s_memtime() - tstart;
                            rec->visits = 0; rec->emits = (uint32_t) sp;
                        }
                        phase = 2;
                    }
                }
            }
        }
        if (__ballot(phase != 2) == 0) break;
        // One iteration: the chains that have left their window look the next 16 positions up (doR), then every chain
        // with a candidate at or after its scan position visits it (doV) — the chains that just looked up included.
        if (SWSEM_SYNC_SPAN && __ballot(phase != 2 && scan - start0 < syncRel) == 0) syncRel += SWSEM_SYNC_SPAN;
        const bool doR = phase != 2 && (scan < wb || scan >= wb + WN4) && (!SWSEM_SYNC_SPAN || scan - start0 < syncRel);

        // ---- doR, issue: K-mer hashes of the window (run_chain_lazy's refill), then the table gather
        int32_t cnt = 0;
        uint32_t hsh = 0;
        ht_entry hte = 0;
        if (doR) {
            wb = scan;
            cnt = p1 - wb < WN4 ? p1 - wb : WN4;
            const int nw = K / 4;
            uint32_t h = (uint32_t) K, f = FP_SEED;
            // the window's bytes: from the chain's query cache in LDS (the 256 bytes around the last visit, or the last
            // chunk fetched here) when they are all there, else from memory — one more round trip, and the cache is
            // refilled from the window on so that the windows that follow find their bytes
            const int32_t last = wb + cnt + K - 1;                 // one past the last byte the window's K-mers need
#ifndef SWSEM_QC_MODE
#define SWSEM_QC_MODE 3
#endif
            if ((SWSEM_QC_MODE & 1) && wb >= qlo && last <= qhi) {
                const uint32_t o = (uint32_t) (wb - qb0) + ((int32_t) gl < cnt ? gl : 0u);   // (lanes past the range repeat lane 0: no reads beyond the window's bytes)
                const uint32_t *w = (const uint32_t *) (qcache[grp] + (o & ~3u));
                const uint32_t sft = o & 3u;
                uint32_t lo = w[0];
                for (int x = 0; x < nw; x++) {
                    const uint32_t hi = w[x + 1];
                    const uint32_t wd = __builtin_amdgcn_alignbyte(hi, lo, sft);
                    h = hash_step(h, wd, (uint32_t) x);
                    f = fp_step(f, wd);
                    lo = hi;
                }
            } else {
                const uintptr_t A = (uintptr_t) (q + wb);
                const uint32_t sh = (uint32_t) (A & 3);
                const uint32_t *A0 = (const uint32_t *) (A & ~(uintptr_t) 3);
                const uint32_t ndw = (sh + (uint32_t) cnt + (uint32_t) K - 1u + 3u) >> 2;     // <= 16 (K <= K_MAX4)
                const uint32_t a = gl < ndw ? A0[gl] : 0u;
                const bool whole = (uint32_t) wb + 16u * gl + 16u <= n;     // this lane's 16-byte chunk of [wb, wb + 256) lies inside the contig
                uint4 chunk = make_uint4(0, 0, 0, 0);
                if (whole) chunk = ld_u128(q + wb + 16 * (int32_t) gl);
                const uint32_t o = sh + gl, wi = gbase + (o >> 2), sft = o & 3u;
                uint32_t lo = gread(a, wi);
                for (int x = 0; x < nw; x++) {
                    const uint32_t hi = gread(a, wi + (uint32_t) x + 1u);
                    const uint32_t wd = __builtin_amdgcn_alignbyte(hi, lo, sft);
                    h = hash_step(h, wd, (uint32_t) x);
                    f = fp_step(f, wd);
                    lo = hi;
                }
                *(uint4 *) (qcache[grp] + 16u * gl) = chunk;
                qb0 = wb; qlo = wb;
                const uint32_t nwhole = (n - (uint32_t) wb) >> 4;
                qhi = wb + 16 * (int32_t) (nwhole < (uint32_t) GL ? nwhole : (uint32_t) GL);
            }
            hsh = f;                                               // (what the consume step needs of the hashes: the fingerprint's)
            if ((int32_t) gl < cnt) hte = v.ht[h & v.mask];
        }

        // ---- doR, consume: window test and fingerprint per position, the group's candidate lanes
        if (doR) {
            uint32_t e = 0;
            if ((int32_t) gl < cnt) {
                e = ht_value<LAPS>(v, hte, hsh);
                if (e != 0) {
                    uint64_t wlo, whi;
                    if (!window_ok(v, lock, (uint64_t) e << v.k1ord, wlo, whi)) e = 0;
                    // a stale entry — older than the text at the slot it points at, whatever its fingerprint (the same K-mer
                    // once stood there: collections share most of theirs) — whose slot was sampled again by the load that
                    // wrote the present text cannot verify (lap_want): two bytes of the slot's tag here instead of a visit's
                    // iteration and its 256 bytes
                    else if (stale_settled<LAPS>(v, hte, e)) e = 0;
                }
            }
            went = e;
            m16 = gballot(e != 0, gbase);
        }

        // ---- first candidate at or after the scan position; a window without one is left behind
        uint32_t mk = 0;
        if (phase != 2 && scan >= wb && scan < wb + WN4) {              // (a chain waiting for the others stands outside its window)
            mk = m16 & (0xFFFFu << (uint32_t) (scan - wb)) & 0xFFFFu;
            if (!mk) scan = wb + WN4 < p1 ? wb + WN4 : p1;
        }
        const bool doV = mk != 0;

        // ---- doV, issue: the 256 bytes around the candidate, reference and query (visit())
        int32_t i = 0, limL = 0, limR = 0;
        uint64_t c = 0, dlo = 0;
        const uint8_t *r0 = v.ref, *q0 = q;
        bool full = false;
        uint4 xr = make_uint4(0, 0, 0, 0), xq = make_uint4(0, 0, 0, 0);
        const int32_t rel = 16 * (int32_t) gl - LEFT4;             // first byte of the lane's chunk, relative to the candidate
        if (doV) {
            const uint32_t l = (uint32_t) __builtin_ctz(mk);
            i = wb + (int32_t) l;
            const uint32_t val = gread(went, gbase + l);
            c = (uint64_t) val << v.k1ord;                         // htDecodePos, .h:133
            uint64_t lo = 0, hi = 0;
            window_ok(v, lock, c, lo, hi);
            r0 = v.ref + c; q0 = q + i;
            const uint64_t ra = hi - (c + (uint64_t) K);
            const uint32_t rbq = n - (uint32_t) (i + K);
            limR = (int32_t) (ra < (uint64_t) rbq ? (uint32_t) ra : rbq);
            dlo = c - lo;
            limL = (uint64_t) i < dlo ? i : (int32_t) dlo;
            full = gl < (uint32_t) LL4 ? -rel <= limL : (gl < (uint32_t) NL4 && rel + 16 <= K + limR);
            const bool qwhole = i + rel >= 0 && (uint32_t) (i + rel) + 16u <= n;   // the query side of the chunk lies inside the contig
            if (full) xr = ld_u128(r0 + rel);
            if (qwhole) xq = ld_u128(q0 + rel);
        }

        if (doV) {
            // ---- visit(): verify the K-mer, first step of the right and of the left run
            {                                                      // the query bytes just read become the chain's query cache
                if (SWSEM_QC_MODE & 2) {
                *(uint4 *) (qcache[grp] + 16u * gl) = xq;
                qb0 = i - LEFT4;
                const int32_t lowest = qb0 < 0 ? qb0 + 16 * ((-qb0 + 15) / 16) : qb0;
                const int32_t room = (int32_t) n - qb0;
                qlo = lowest;
                qhi = qb0 + 16 * (room >= 16 * GL ? GL : room / 16);
                }
            }
            const uint4 x = full ? xor128(xr, xq) : make_uint4(0, 0, 0, 0);
            const bool stop = !full || nz128(x);
            const uint32_t eqF = eq_low(x), eqL = eq_high(x);
            const uint32_t stop16 = gballot(stop, gbase) & ((1u << NL4) - 1u), full16 = gballot(full, gbase);
            const uint32_t fwd = stop16 >> LL4;
            int32_t fwdlen = 16 * (NL4 - LL4);
            bool capR = fwd == 0, capL = false;
            if (!capR) {
                const uint32_t lf = (uint32_t) __builtin_ctz(fwd);
                fwdlen = 16 * (int32_t) lf;
                if ((full16 >> (LL4 + lf)) & 1u) fwdlen += (int32_t) gread(eqF, gbase + LL4 + lf);
                else { const int32_t lim = K + limR; while (fwdlen < lim && r0[fwdlen] == q0[fwdlen]) fwdlen++; }   // the lane that holds the limit
            }
            if (fwdlen < K) scan = i + 1;                                     // memcmp(curr1, curr2, K) fails, .cpp:298: no effect
            else {
                int32_t rext = fwdlen - K;
                const uint32_t bwd = stop16 & ((1u << LL4) - 1u);
                int32_t ell = LEFT4;
                capL = bwd == 0;
                if (!capL) {
                    const uint32_t lb = 31u - (uint32_t) __builtin_clz(bwd);  // the stopping lane nearest to the candidate
                    ell = 16 * (LL4 - 1 - (int32_t) lb);
                    if ((full16 >> lb) & 1u) ell += (int32_t) gread(eqL, gbase + lb);
                    else while (ell < limL && r0[-ell - 1] == q0[-ell - 1]) ell++;
                }
                // ---- process_hit(): .cpp:250-315 on the chain's match stack
                const uint64_t d = dlo;
                const int32_t loDist = (int32_t) (d > 0x7FFFFFFFull ? 0x7FFFFFFFull : d);
                auto need_ell = [&](int32_t want) {
                    if (capL && want > ell) {
                        const uint32_t jmax = (uint64_t) i < d ? (uint32_t) i : (uint32_t) d;
                        ell = (int32_t) group_lcp_bwd(r0, q0, (uint32_t) ell, jmax, gl, gbase);
                        capL = false;
                    }
                };
                auto stk_get = [&](int idx, int32_t &mPos, int32_t &mLen) {
                    if (idx >= ringLow) { const uint4 e = ring[grp][idx & (RING4 - 1)]; mPos = (int32_t) e.x; mLen = (int32_t) e.y; }
                    else { mPos = (int32_t) (uint32_t) st[idx].posDest; mLen = (int32_t) (uint32_t) st[idx].len; }
                };
                int32_t s = 0;                                                // (p1, p2) = (c - s, i - s)
                int keep = sp;                                                // resSizeWithoutOverlapped
                bool brokeOut = false;
                while (keep-- > 0) {                                          // .cpp:253
                    int32_t mPos, mLen;
                    stk_get(keep, mPos, mLen);
                    const int32_t mEnd = mPos + mLen;
                    if (mEnd < i - s) {                                       // .cpp:255
                        int32_t t = loDist < i - mEnd + 1 ? loDist : i - mEnd + 1;
                        need_ell(t - 1);
                        if (ell + 1 < t) t = ell + 1;
                        if (t > s) s = t;                                     // .cpp:257-259
                        if (i - s > mEnd - 1) { brokeOut = true; break; }     // .cpp:260-261
                        s -= 1;                                               // .cpp:262
                    }
                    const int32_t dl = (i - s) - mPos;                        // lastDelta, .cpp:264
                    bool fail = (loDist - s < dl) || (mLen > OVERLAP_MATCH_MAX_LENGTH);
                    if (!fail) {
                        need_ell(s + dl);
                        fail = ell < s + dl;                                  // strcmplcp(...) != 0, .cpp:266
                    }
                    if (fail) { s += 1; brokeOut = true; break; }             // .cpp:267-268
                    s += dl;                                                  // .cpp:270
                }
                if (!brokeOut) keep = -1;
                if (keep < minTouched) minTouched = keep;
                if (keep < 0) {                                               // .cpp:277-280
                    int32_t t = loDist < i + 1 ? loDist : i + 1;
                    need_ell(t - 1);
                    if (ell + 1 < t) t = ell + 1;
                    if (t > s) s = t;
                } else {                                                      // .cpp:285-289
                    int32_t mPos, mLen;
                    stk_get(keep, mPos, mLen);
                    const int32_t overlap = (mPos + mLen) - (i - s + 1);
                    if (overlap > 0) s -= overlap;
                }
                ++keep;
                bool emitted = false;
                if (K + rext + s > (int32_t) v.minLen || capR) {              // right1 - p1 > minMatchLength, .cpp:298
                    if (capR) {
                        rext = (int32_t) group_lcp_fwd(r0 + K, q0 + K, (uint32_t) rext, (uint32_t) limR, gl, gbase);
                        capR = false;
                    }
                    if (K + rext + s > (int32_t) v.minLen) {
                        const uint64_t mSrc = c - (uint64_t) (int64_t) (s - 1);
                        const uint32_t mLen = (uint32_t) (K + rext + s - 1), mDest = (uint32_t) (i - s + 1);
                        int32_t skip = K + rext;                              // (matchEnd - i2), k2 == 1, .cpp:308
                        skip -= skip > v.skipMargin ? v.skipMargin : skip;
                        scan = skip ? i + skip : i + 1;                       // .cpp:310-313 then the loop's i2 += k2
                        // resMatches.resize(keep); resMatches.push_back(m)   (.cpp:299-300)
                        sp = keep;
                        if (gl == 0) {
                            Row r;
                            r.posSrc = mSrc; r.len = mLen; r.posDest = mDest; r.scanAfter = scan;
                            st[sp] = r;
                            ring[grp][sp & (RING4 - 1)] = make_uint4(mDest, mLen, (uint32_t) mSrc, (uint32_t) (mSrc >> 32));
                        }
                        if (sp - (RING4 - 1) > ringLow) ringLow = sp - (RING4 - 1);
                        if (ringLow > sp) ringLow = sp;
                        sp++;
                        if (keep < minKeep) minKeep = keep;
                        emitted = true;
                    }
                }
                if (!emitted) scan = i + 1;
            }
        }
    }
}

}  // namespace swk
