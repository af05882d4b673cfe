// Input stage on the device (include/mbgc_fasta.h): kseq_read_lossless_fasta (utils/kseq.h:233-274) for whole files
// resident in HBM. The reader is a byte-serial state machine, but its state at a byte is tiny — "is the line
// this byte belongs to a header line" and where that line started — and it is decided by the nearest '\n'
// before the byte. So: (1) every 4096-byte chunk summarises itself independently of what comes before
// (its first/last newline, what it keeps after its first newline, its header starts), (2) one wave per file
// scans the chunk summaries (a chunk with a newline fixes the state for what follows, one without passes it
// on), (3) every chunk, now knowing the state it starts in and how many bytes / records precede it, writes its
// sequence bytes and its records and votes on the line-length rule. Streaming work: the file is read twice,
// the sequences are written once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <sys/mman.h>

#include "../../include/mbgc_fasta.h"

namespace fa {

constexpr int CHUNK = 4096, THREADS = 256, PER = CHUNK / THREADS;    // 16 bytes per thread
constexpr int WAVE = 64;

struct FileDesc {
    uint64_t off, n;          // bytes of the file in the input buffer
    uint32_t chunk0, nchunks;
};

struct ChunkSum {             // what a chunk knows on its own
    int32_t firstNL, lastNL;  // offsets inside the chunk, -1: none
    uint32_t keepAfter;       // bytes kept among those after firstNL (their lines start inside the chunk)
    uint32_t hdrStarts;       // header lines starting in the chunk (a '>' on the file's last byte is not one, kseq.h:243)
    uint8_t fresh;            // the chunk's first byte starts a line (file start, or the byte before it is '\n')
    uint8_t firstGt;          // the chunk's first byte is '>'
    uint8_t openHdr;          // the line open at the chunk's end is a header line (valid when lastNL >= 0 and it is not the last byte)
    uint8_t pad;
};

struct ChunkIn {              // what the scan adds
    uint64_t lineStart;       // file offset where the line open at the chunk's first byte started
    uint64_t keptBefore;      // sequence bytes of the file before the chunk
    uint32_t recBefore;       // records of the file before the chunk
    uint32_t inHdr;           // that open line is a header line
};

struct FileOut {              // per file, written by the scan and the emit kernels
    uint64_t kept, recs;
    unsigned long long minLine, maxLine, maxLast;   // lengths of the non-last lines of all records / of their last lines
    uint32_t emptyLine, firstNotGt;
};

__device__ __forceinline__ uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t) (c - 32) : c; }

// stage the chunk in LDS; returns its length
__device__ __forceinline__ uint32_t stage(const uint8_t *__restrict__ f, const FileDesc &fd, uint32_t c, uint8_t *lds) {
    const uint64_t cs = (uint64_t) c * CHUNK;
    const uint32_t len = (uint32_t) (fd.n - cs < CHUNK ? fd.n - cs : CHUNK);
    const uint8_t *src = f + fd.off + cs;
    const uint32_t o = threadIdx.x * PER;
    if (o + PER <= len) {
        uint4 t;
        memcpy(&t, src + o, PER);
        *(uint4 *) (lds + o) = t;
    } else
        for (uint32_t k = o; k < len && k < o + PER; k++) lds[k] = src[k];
    __syncthreads();
    return len;
}

// last '\n' at or before every thread's segment start (exclusive of the segment), as an offset in the chunk, -1: none.
// `mine` = the thread's own last newline offset or -1. lds2: THREADS ints.
__device__ __forceinline__ int32_t prev_newline(int32_t mine, int32_t *lds2) {
    lds2[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < THREADS; d <<= 1) {                           // inclusive max-scan (Hillis-Steele)
        const int32_t other = threadIdx.x >= (unsigned) d ? lds2[threadIdx.x - d] : -1;
        __syncthreads();
        if (other > lds2[threadIdx.x]) lds2[threadIdx.x] = other;
        __syncthreads();
    }
    const int32_t r = threadIdx.x ? lds2[threadIdx.x - 1] : -1;
    __syncthreads();
    return r;
}

template <int NT>
__device__ uint32_t block_sum_scan(uint32_t x, uint32_t *lds, uint32_t *total) {     // exclusive scan over the block
    const uint32_t lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint32_t inc = x;
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t y = (uint32_t) __shfl_up((int) inc, d);
        if ((int) lane >= d) inc += y;
    }
    __syncthreads();
    if (lane == WAVE - 1) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < NT / WAVE; i++) { const uint32_t t = lds[i]; lds[i] = run; run += t; }
        lds[NT / WAVE] = run;
    }
    __syncthreads();
    *total = lds[NT / WAVE];
    const uint32_t r = lds[w] + inc - x;
    __syncthreads();
    return r;
}

// (1) per chunk
__global__ void __launch_bounds__(THREADS) k_fa_summary(const uint8_t *__restrict__ f, const FileDesc *__restrict__ files,
                                                        const uint32_t *__restrict__ owner, ChunkSum *__restrict__ sums) {
    __shared__ uint8_t ch[CHUNK + 16];
    __shared__ int32_t nl[THREADS];
    __shared__ uint32_t red[THREADS / WAVE + 2];
    __shared__ int32_t firstNL, lastNL;
    const FileDesc fd = files[owner[blockIdx.x]];
    const uint32_t c = blockIdx.x - fd.chunk0;
    const uint32_t len = stage(f, fd, c, ch);
    const uint64_t cs = (uint64_t) c * CHUNK;
    if (threadIdx.x == 0) { firstNL = 0x7fffffff; lastNL = -1; }
    __syncthreads();
    const uint32_t o = threadIdx.x * PER;
    int32_t myFirst = 0x7fffffff, myLast = -1;
    for (uint32_t k = o; k < o + PER && k < len; k++)
        if (ch[k] == '\n') { if (myFirst == 0x7fffffff) myFirst = (int32_t) k; myLast = (int32_t) k; }
    if (myLast >= 0) { atomicMin(&firstNL, myFirst); atomicMax(&lastNL, myLast); }
    const int32_t prevNL = prev_newline(myLast, nl);
    __syncthreads();
    const int32_t fNL = firstNL == 0x7fffffff ? -1 : firstNL;
    // walk the segment: bytes after the chunk's first newline know their line start
    uint32_t keep = 0, hdr = 0;
    int32_t ls = prevNL >= 0 ? prevNL + 1 : -1;                       // start of the open line if it lies in the chunk
    bool isHdr = ls >= 0 && ls < (int32_t) len && ch[ls] == '>';
    const bool fresh = cs == 0 || f[fd.off + cs - 1] == '\n';
    for (uint32_t k = o; k < o + PER && k < len; k++) {
        const uint8_t b = ch[k];
        const bool lineStart = k == 0 ? fresh : ch[k - 1] == '\n';
        if (lineStart) { ls = (int32_t) k; isHdr = b == '>'; if (isHdr && cs + k + 1 < fd.n) hdr++; }
        if (fNL >= 0 && (int32_t) k > fNL && !isHdr && b != '\n') keep++;
    }
    uint32_t tk, th;
    block_sum_scan<THREADS>(keep, red, &tk);
    block_sum_scan<THREADS>(hdr, red, &th);
    if (threadIdx.x == 0) {
        ChunkSum s;
        s.firstNL = fNL; s.lastNL = lastNL; s.keepAfter = tk; s.hdrStarts = th;
        s.fresh = fresh; s.firstGt = len && ch[0] == '>';
        s.openHdr = lastNL >= 0 && lastNL + 1 < (int32_t) len && ch[lastNL + 1] == '>';
        s.pad = 0;
        sums[blockIdx.x] = s;
    }
}

// (2) one wave per file
__global__ void __launch_bounds__(WAVE) k_fa_scan(const uint8_t *__restrict__ f, const FileDesc *__restrict__ files,
                                                  const ChunkSum *__restrict__ sums, ChunkIn *__restrict__ ins, FileOut *__restrict__ fout) {
    const FileDesc fd = files[blockIdx.x];
    const int lane = threadIdx.x;
    // state carried across groups of 64 chunks
    bool carryHdr = fd.n && f[fd.off] == '>';                         // the file's first line (no newline seen yet)
    uint64_t carryLS = 0, kept = 0;
    uint32_t recs = 0;
    for (uint32_t g0 = 0; g0 < fd.nchunks; g0 += WAVE) {
        const uint32_t c = g0 + lane;
        const bool live = c < fd.nchunks;
        ChunkSum s;
        s.firstNL = -1; s.lastNL = -1; s.keepAfter = 0; s.hdrStarts = 0; s.fresh = 0; s.firstGt = 0; s.openHdr = 0; s.pad = 0;
        if (live) s = sums[fd.chunk0 + c];
        const uint64_t cs = (uint64_t) c * CHUNK;
        const uint32_t len = live ? (uint32_t) (fd.n - cs < CHUNK ? fd.n - cs : CHUNK) : 0;
        // nearest earlier chunk of the group that fixes the state of what follows: one with a newline (the line open
        // at its end), or one whose first byte starts a line and that has no newline (that very line)
        int idx = (live && (s.lastNL >= 0 || s.fresh)) ? lane : -1;
        for (int d = 1; d < WAVE; d <<= 1) {
            const int y = __shfl_up(idx, d);
            if (lane >= d && y > idx) idx = y;
        }
        const int prev = __shfl_up(idx, 1);
        const int pidx = lane ? prev : -1;
        // state a chunk with a newline leaves behind: header flag and start of its open line
        // (a chunk that ends with its newline leaves a line that starts with the next chunk: that chunk is "fresh" and
        // never asks)
        const uint64_t myLS = s.lastNL >= 0 ? cs + (uint64_t) (s.lastNL + 1) : cs;
        const int myHdr = s.lastNL >= 0 ? (int) s.openHdr : (int) s.firstGt;
        const int srcLane = pidx >= 0 ? pidx : 0;
        const int pHdr = __shfl(myHdr, srcLane);
        const uint64_t pLS = ((uint64_t) (uint32_t) __shfl((int) (myLS >> 32), srcLane) << 32) | (uint32_t) __shfl((int) (uint32_t) myLS, srcLane);
        bool inHdr; uint64_t inLS;
        if (s.fresh) { inHdr = s.firstGt; inLS = cs; }
        else if (pidx >= 0) { inHdr = pHdr; inLS = pLS; }
        else { inHdr = carryHdr; inLS = carryLS; }
        const uint32_t prefix = s.firstNL >= 0 ? (uint32_t) s.firstNL : len;        // bytes of the open line inside the chunk (no newline among them)
        const uint32_t mine = live ? s.keepAfter + (inHdr ? 0u : prefix) : 0u;
        uint32_t incK = mine, incR = live ? s.hdrStarts : 0u;
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t a = (uint32_t) __shfl_up((int) incK, d), b = (uint32_t) __shfl_up((int) incR, d);
            if (lane >= d) { incK += a; incR += b; }
        }
        if (live) {
            ChunkIn in;
            in.lineStart = inLS; in.keptBefore = kept + incK - mine; in.recBefore = recs + incR - s.hdrStarts; in.inHdr = inHdr;
            ins[fd.chunk0 + c] = in;
        }
        // carry out of the group: the last chunk with a newline (or the old carry), totals
        const int lastIdx = __shfl(idx, WAVE - 1);
        if (lastIdx >= 0) {
            const int lh = __shfl(myHdr, lastIdx);
            const uint64_t ll = ((uint64_t) (uint32_t) __shfl((int) (myLS >> 32), lastIdx) << 32) | (uint32_t) __shfl((int) (uint32_t) myLS, lastIdx);
            carryHdr = lh; carryLS = ll;
        }
        kept += (uint64_t) (uint32_t) __shfl((int) incK, WAVE - 1);
        recs += (uint32_t) __shfl((int) incR, WAVE - 1);
    }
    if (lane == 0) {
        FileOut o;
        o.kept = kept; o.recs = recs; o.minLine = ~0ull; o.maxLine = 0; o.maxLast = 0; o.emptyLine = 0;
        o.firstNotGt = fd.n && f[fd.off] != '>';
        fout[blockIdx.x] = o;
    }
}

// (3) per chunk again
__global__ void __launch_bounds__(THREADS) k_fa_emit(const uint8_t *__restrict__ f, const FileDesc *__restrict__ files,
                                                     const uint32_t *__restrict__ owner, const ChunkSum *__restrict__ sums,
                                                     const ChunkIn *__restrict__ ins, const uint64_t *__restrict__ seqBase,
                                                     const uint64_t *__restrict__ recBase, int uppercase, uint8_t *__restrict__ out,
                                                     mbgc_fasta_record_t *__restrict__ recs, FileOut *__restrict__ fout) {
    __shared__ uint8_t ch[CHUNK + 16];
    __shared__ int32_t nl[THREADS];
    __shared__ uint32_t red[THREADS / WAVE + 2];
    const uint32_t fi = owner[blockIdx.x];
    const FileDesc fd = files[fi];
    const uint32_t c = blockIdx.x - fd.chunk0;
    const uint32_t len = stage(f, fd, c, ch);
    const uint64_t cs = (uint64_t) c * CHUNK;
    const ChunkIn in = ins[blockIdx.x];
    const ChunkSum sm = sums[blockIdx.x];
    const uint32_t o = threadIdx.x * PER;
    int32_t myLast = -1;
    for (uint32_t k = o; k < o + PER && k < len; k++) if (ch[k] == '\n') myLast = (int32_t) k;
    const int32_t prevNL = prev_newline(myLast, nl);
    // state at the thread's first byte
    uint64_t ls = prevNL >= 0 ? cs + (uint64_t) prevNL + 1 : in.lineStart;        // file offset of the open line's start
    bool isHdr = prevNL >= 0 ? (prevNL + 1 < (int32_t) len ? ch[prevNL + 1] == '>' : false) : in.inHdr != 0;
    // pass A: count what the thread keeps and the header starts before each byte
    uint32_t keep = 0, hdr = 0;
    {
        uint64_t l2 = ls; bool h2 = isHdr;
        for (uint32_t k = o; k < o + PER && k < len; k++) {
            const uint8_t b = ch[k];
            const bool lineStart = k == 0 ? sm.fresh : ch[k - 1] == '\n';
            if (lineStart) { l2 = cs + k; h2 = b == '>'; if (h2 && cs + k + 1 < fd.n) hdr++; }
            if (!h2 && b != '\n') keep++;
        }
    }
    uint32_t tk, th;
    const uint32_t keepEx = block_sum_scan<THREADS>(keep, red, &tk);
    const uint32_t hdrEx = block_sum_scan<THREADS>(hdr, red, &th);
    // pass B: the kept bytes are compacted in LDS first (then stored coalesced), records and line lengths on the way
    __shared__ uint8_t packed[CHUNK];
    __shared__ unsigned long long sMin, sMax, sLast;
    __shared__ uint32_t sEmpty;
    if (threadIdx.x == 0) { sMin = ~0ull; sMax = 0; sLast = 0; sEmpty = 0; }
    __syncthreads();
    uint8_t *dst = packed + keepEx;
    mbgc_fasta_record_t *R = recs + recBase[fi];
    uint32_t kk = 0, hh = in.recBefore + hdrEx;                     // hh: records started before the current byte
    unsigned long long mn = ~0ull, mx = 0, mxLast = 0;
    uint32_t empty = 0;
    for (uint32_t k = o; k < o + PER && k < len; k++) {
        const uint8_t b = ch[k];
        const uint64_t p = cs + k;
        const bool lineStart = k == 0 ? sm.fresh : ch[k - 1] == '\n';
        if (lineStart) {
            ls = p; isHdr = b == '>';
            if (isHdr && p + 1 < fd.n) {                            // a record starts here
                R[hh].headerOff = p + 1;
                R[hh].seqOff = in.keptBefore + keepEx + kk;
                hh++;
            }
        }
        const bool term = b == '\n';
        const bool eofTerm = !term && p + 1 == fd.n;                // the file's last line has no newline
        if (term || eofTerm) {
            const uint64_t end = term ? p : p + 1;
            if (isHdr) {
                if (ls + 1 < fd.n && hh > 0) R[hh - 1].headerLen = end - (ls + 1);  // the record whose header line ends here
            } else {
                const unsigned long long d = end - ls;
                const bool lastOfRecord = end + 1 >= fd.n || (term && f[fd.off + p + 1] == '>') || eofTerm;
                if (d == 0) empty = 1;                               // an empty line (kseq.h:251)
                else if (lastOfRecord) { if (d > mxLast) mxLast = d; }
                else { if (d < mn) mn = d; if (d > mx) mx = d; }
            }
        }
        if (!isHdr && !term) dst[kk++] = uppercase ? up(b) : b;
    }
    // one vote per chunk on the line-length rule (a million lines voting one by one serialise on the file's counters)
    if (mn != ~0ull) { atomicMin(&sMin, mn); atomicMax(&sMax, mx); }
    if (mxLast) atomicMax(&sLast, mxLast);
    if (empty) atomicOr(&sEmpty, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sMin != ~0ull) { atomicMin(&fout[fi].minLine, sMin); atomicMax(&fout[fi].maxLine, sMax); }
        if (sLast) atomicMax(&fout[fi].maxLast, sLast);
        if (sEmpty) atomicOr(&fout[fi].emptyLine, 1u);
    }
    uint8_t *g = out + seqBase[fi] + in.keptBefore;
    for (uint32_t k = threadIdx.x * 16; k < tk; k += THREADS * 16) {
        if (k + 16 <= tk) {
            uint4 t = *(const uint4 *) (packed + k);
            memcpy(g + k, &t, 16);
        } else
            for (uint32_t j = k; j < tk; j++) g[j] = packed[j];
    }
}

std::string g_err;
int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define FCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fa::fail(-100, "%s: %s", #x, hipGetErrorString(e_)); } while (0)

template <class T>
struct Buf {
    T *p = nullptr; size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void) hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 64;
        if (hipMalloc((void **) &p, want * sizeof(T)) != hipSuccess) return fail(-101, "device allocation of %zu bytes failed", want * sizeof(T));
        cap = want;
        return 0;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace fa

struct mbgc_fasta {
    int device = 0;
    hipStream_t stream = nullptr;
    fa::Buf<fa::FileDesc> dFiles;
    fa::Buf<uint32_t> dOwner;
    fa::Buf<fa::ChunkSum> dSums;
    fa::Buf<fa::ChunkIn> dIns;
    fa::Buf<fa::FileOut> dOut;
    fa::Buf<uint64_t> dBases;
    fa::Buf<mbgc_fasta_record_t> dRecs;
    fa::Buf<uint8_t> dHostIn, dHostOut;         // mbgc_fasta_parse_host: the file and its sequences on the device
};

extern "C" {

const char *mbgc_fasta_last_error(void) { return fa::g_err.c_str(); }

int mbgc_fasta_create(mbgc_fasta_t **out, int device) {
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev)
        return fa::fail(-102, "no HIP device %d (the input stage has no CPU fallback)", device);
    FCHK(hipSetDevice(device));
    mbgc_fasta *p = new mbgc_fasta();
    p->device = device;
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { delete p; return fa::fail(-100, "hipStreamCreate failed"); }
    *out = p;
    return 0;
}

void mbgc_fasta_destroy(mbgc_fasta_t *p) {
    if (!p) return;
    (void) hipSetDevice(p->device);
    if (p->stream) { (void) hipStreamSynchronize(p->stream); (void) hipStreamDestroy(p->stream); }
    p->dFiles.release(); p->dOwner.release(); p->dSums.release(); p->dIns.release(); p->dOut.release(); p->dBases.release(); p->dRecs.release(); p->dHostIn.release(); p->dHostOut.release();
    delete p;
}

int mbgc_fasta_parse_batch_dev(mbgc_fasta_t *p, const uint8_t *files_dev, const uint64_t *fileOff, int nf, int uppercaseDNA,
                               uint8_t *seq_out_dev, uint64_t outCap, uint64_t *seqBase,
                               mbgc_fasta_record_t *records, uint64_t recCap, uint64_t *recBase,
                               uint64_t *dnaLineLen, int *status) {
    using namespace fa;
    if (nf <= 0) return fail(-103, "empty batch");
    FCHK(hipSetDevice(p->device));
    std::vector<FileDesc> files(nf);
    std::vector<uint32_t> owner;
    uint32_t chunks = 0;
    for (int i = 0; i < nf; i++) {
        if (fileOff[i + 1] < fileOff[i]) return fail(-103, "file offsets must ascend");
        files[i].off = fileOff[i]; files[i].n = fileOff[i + 1] - fileOff[i];
        files[i].chunk0 = chunks;
        files[i].nchunks = (uint32_t) ((files[i].n + CHUNK - 1) / CHUNK);
        owner.insert(owner.end(), files[i].nchunks, (uint32_t) i);
        chunks += files[i].nchunks;
    }
    int r;
    if ((r = p->dFiles.reserve(nf)) || (r = p->dOwner.reserve(std::max<uint32_t>(chunks, 1))) || (r = p->dSums.reserve(std::max<uint32_t>(chunks, 1))) ||
        (r = p->dIns.reserve(std::max<uint32_t>(chunks, 1))) || (r = p->dOut.reserve(nf)) || (r = p->dBases.reserve(2 * (size_t) nf + 2)))
        return r;
    FCHK(hipMemcpyAsync(p->dFiles.p, files.data(), nf * sizeof(FileDesc), hipMemcpyHostToDevice, p->stream));
    if (chunks) FCHK(hipMemcpyAsync(p->dOwner.p, owner.data(), chunks * sizeof(uint32_t), hipMemcpyHostToDevice, p->stream));
    if (chunks) k_fa_summary<<<dim3(chunks), dim3(THREADS), 0, p->stream>>>(files_dev, p->dFiles.p, p->dOwner.p, p->dSums.p);
    k_fa_scan<<<dim3(nf), dim3(WAVE), 0, p->stream>>>(files_dev, p->dFiles.p, p->dSums.p, p->dIns.p, p->dOut.p);
    FCHK(hipGetLastError());
    std::vector<FileOut> fo(nf);
    FCHK(hipMemcpyAsync(fo.data(), p->dOut.p, nf * sizeof(FileOut), hipMemcpyDeviceToHost, p->stream));
    FCHK(hipStreamSynchronize(p->stream));
    std::vector<uint64_t> bases(2 * (size_t) nf + 2);
    uint64_t *sb = bases.data(), *rb = bases.data() + nf + 1;
    sb[0] = 0; rb[0] = 0;
    for (int i = 0; i < nf; i++) { sb[i + 1] = sb[i] + fo[i].kept; rb[i + 1] = rb[i] + fo[i].recs; }
    if (sb[nf] > outCap) return fail(-104, "sequence output needs %llu bytes, capacity %llu", (unsigned long long) sb[nf], (unsigned long long) outCap);
    if (rb[nf] > recCap) { recBase[nf] = rb[nf]; return fail(-104, "record table needs %llu entries, capacity %llu", (unsigned long long) rb[nf], (unsigned long long) recCap); }
    if ((r = p->dRecs.reserve(std::max<uint64_t>(rb[nf], 1)))) return r;
    FCHK(hipMemcpyAsync(p->dBases.p, bases.data(), bases.size() * sizeof(uint64_t), hipMemcpyHostToDevice, p->stream));
    if (chunks) k_fa_emit<<<dim3(chunks), dim3(THREADS), 0, p->stream>>>(files_dev, p->dFiles.p, p->dOwner.p, p->dSums.p, p->dIns.p, p->dBases.p,
                                                                         p->dBases.p + nf + 1, uppercaseDNA, seq_out_dev, p->dRecs.p, p->dOut.p);
    FCHK(hipGetLastError());
    if (rb[nf]) FCHK(hipMemcpyAsync(records, p->dRecs.p, rb[nf] * sizeof(mbgc_fasta_record_t), hipMemcpyDeviceToHost, p->stream));
    FCHK(hipMemcpyAsync(fo.data(), p->dOut.p, nf * sizeof(FileOut), hipMemcpyDeviceToHost, p->stream));
    FCHK(hipStreamSynchronize(p->stream));
    for (int i = 0; i < nf; i++) {
        seqBase[i] = sb[i]; recBase[i] = rb[i];
        // a record's contig ends where the next one starts
        for (uint64_t k = rb[i]; k < rb[i + 1]; k++) {
            records[k].seqLen = (k + 1 < rb[i + 1] ? records[k + 1].seqOff : fo[i].kept) - records[k].seqOff;
        }
        // kseq status and KSEQ_DNA_LINE_LENGTH (kseq.h:251-265, MGMP.cpp:12-14)
        const bool haveLine = fo[i].minLine != ~0ull;
        const unsigned long long L = haveLine ? fo[i].minLine : 0;
        int st = MBGC_FASTA_OK;
        if (fo[i].firstNotGt) st = MBGC_FASTA_ENOTFASTA;
        else if (fo[i].emptyLine || (haveLine && fo[i].minLine != fo[i].maxLine) || (haveLine && fo[i].maxLast > L)) st = MBGC_FASTA_ELINES;
        status[i] = st;
        dnaLineLen[i] = st == MBGC_FASTA_OK ? L : 0;
    }
    seqBase[nf] = sb[nf]; recBase[nf] = rb[nf];
    return 0;
}

int mbgc_fasta_parse_host(mbgc_fasta_t *p, const uint8_t *file_host, uint64_t n, int uppercaseDNA, uint8_t *seq_out_host,
                          uint64_t *seqBytes, mbgc_fasta_record_t *records, uint64_t recCap, uint64_t *nrec,
                          uint64_t *dnaLineLen, int *status) {
    using namespace fa;
    FCHK(hipSetDevice(p->device));
    int r;
    if ((r = p->dHostIn.reserve(std::max<uint64_t>(n, 1))) || (r = p->dHostOut.reserve(std::max<uint64_t>(n, 1)))) return r;
    if (n) FCHK(hipMemcpy(p->dHostIn.p, file_host, n, hipMemcpyHostToDevice));
    const uint64_t off[2] = {0, n};
    uint64_t sb[2] = {0, 0}, rb[2] = {0, 0};
    *nrec = 0; *seqBytes = 0;
    if (n == 0) { *status = MBGC_FASTA_OK; *dnaLineLen = 0; return 0; }
    r = mbgc_fasta_parse_batch_dev(p, p->dHostIn.p, off, 1, uppercaseDNA, p->dHostOut.p, n, sb, records, recCap, rb, dnaLineLen, status);
    if (r) { *nrec = rb[1]; return r; }
    *nrec = rb[1]; *seqBytes = sb[1];
    if (sb[1]) FCHK(hipMemcpy(seq_out_host, p->dHostOut.p, sb[1], hipMemcpyDeviceToHost));
    return 0;
}

// Page-locked staging memory. hipHostMalloc takes 33-37 ms for a round's 160 MB on this host (profiles/pin_bench.hip) — twice,
// in front of the first round; memory of the process's own in 2 MB pages (madvise, where the system gives them) is registered in
// about 1 ms once it is there, and touching it takes 10 ms instead of 27. So: mmap, ask for huge pages, register.
namespace fa {
static std::mutex g_hostMu;
static std::map<void *, std::pair<void *, size_t>> g_hostMaps;      // registered pointer -> (mapping, its length)
}

int mbgc_fasta_host_alloc(mbgc_fasta_t *p, uint64_t bytes, void **out) {
    using namespace fa;
    FCHK(hipSetDevice(p->device));
    *out = nullptr;
    const size_t HUGE = (size_t) 2 << 20, n = ((bytes ? bytes : 1) + HUGE - 1) & ~(HUGE - 1);
    void *m = mmap(nullptr, n + HUGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return fail(-101, "cannot map %llu B of host memory", (unsigned long long) bytes);
    void *a = (void *) (((uintptr_t) m + HUGE - 1) & ~(uintptr_t) (HUGE - 1));
    (void) madvise(a, n, MADV_HUGEPAGE);
    if (hipHostRegister(a, n, hipHostRegisterDefault) != hipSuccess) {
        (void) hipGetLastError();
        munmap(m, n + HUGE);
        return fail(-101, "cannot pin %llu B of host memory", (unsigned long long) bytes);
    }
    { std::lock_guard<std::mutex> g(g_hostMu); g_hostMaps[a] = std::make_pair(m, n + HUGE); }
    *out = a;
    return 0;
}

int mbgc_fasta_host_free(mbgc_fasta_t *p, void *ptr) {
    using namespace fa;
    FCHK(hipSetDevice(p->device));
    if (!ptr) return 0;
    std::pair<void *, size_t> m;
    {
        std::lock_guard<std::mutex> g(g_hostMu);
        auto it = g_hostMaps.find(ptr);
        if (it == g_hostMaps.end()) return fail(-100, "mbgc_fasta_host_free: not a pointer of mbgc_fasta_host_alloc");
        m = it->second;
        g_hostMaps.erase(it);
    }
    FCHK(hipHostUnregister(ptr));
    munmap(m.first, m.second);
    return 0;
}

int mbgc_fasta_upload(mbgc_fasta_t *p, uint8_t *dst_dev, const void *src_host, uint64_t bytes) {
    using namespace fa;
    FCHK(hipSetDevice(p->device));
    if (bytes) FCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, p->stream));
    FCHK(hipStreamSynchronize(p->stream));
    return 0;
}

}  // extern "C"
