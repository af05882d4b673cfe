// TEST INFRASTRUCTURE ONLY (oracle/_ref). The reference's own FASTA reader — utils/kseq.h, included from
// where it lies under /root/reference — instantiated over a memory buffer instead of mgmpInFile, and driven
// the way MultipleGenomeMatchingProcessor::processTarget drives it (matching/MultipleGenomeMatchingProcessor.cpp
// :359-372, KSEQ_READ :9-10 with enableDNALineLengthDetection and no lossy parsing, KSEQ_DNA_LINE_LENGTH :12-14).
// It pins oracle/fasta_oracle.c (tests/test_fasta_input.py).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <cstdio>
#include "utils/kseq.h"

namespace {
struct MemFile { const unsigned char *p; uint64_t n, pos; };
int64_t mem_read(MemFile *f, void *buf, int64_t sz) {
    const uint64_t left = f->n - f->pos;
    const uint64_t k = left < (uint64_t) sz ? left : (uint64_t) sz;
    memcpy(buf, f->p + f->pos, k);
    f->pos += k;
    return (int64_t) k;
}
KSEQ_INIT(MemFile *, mem_read)
}

extern "C" {

struct reff_record { uint64_t headerLen, seqOff, seqLen; };

// returns the number of records; headers are appended to hdrOut (back to back), sequences to seqOut
uint64_t reff_parse(const unsigned char *file, uint64_t n, int uppercase, unsigned char *seqOut, unsigned char *hdrOut,
                    reff_record *rec, uint64_t recCap, uint64_t *seqBytes, uint64_t *dnaLineLen, int *status) {
    MemFile mf = {file, n, 0};
    kseq_t *seq = kseq_init(&mf);
    uint64_t nrec = 0, out = 0, hout = 0;
    int64_t st;
    while ((st = kseq_read_lossless_fasta(seq)) >= 0) {
        if (uppercase) for (uint64_t i = 0; i < seq->seq.l; i++) seq->seq.s[i] = (char) toupper(seq->seq.s[i]);   // PgHelpers::upperSequence
        if (nrec < recCap) { rec[nrec].headerLen = seq->name.l; rec[nrec].seqOff = out; rec[nrec].seqLen = seq->seq.l; }
        memcpy(hdrOut + hout, seq->name.s, seq->name.l); hout += seq->name.l;
        memcpy(seqOut + out, seq->seq.s, seq->seq.l); out += seq->seq.l;
        nrec++;
    }
    *status = st == -1 ? 0 : (int) st;
    *seqBytes = out;
    *dnaLineLen = seq->dnaLineLen == (uint64_t) DNA_NOT_WELLFORMED ? 0 : seq->dnaLineLen;
    kseq_destroy(seq);
    return nrec;
}

}
