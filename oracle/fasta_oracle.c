/* TEST INFRASTRUCTURE ONLY — CPU restatement of the input stage of the compress path:
 * kseq_read_lossless_fasta (utils/kseq.h:233-274) driven over a whole file the way processTarget does
 * (matching/MultipleGenomeMatchingProcessor.cpp:359-372: while ((status = KSEQ_READ(seq)) >= 0) ...,
 * then validate_kseq_status :16-35 and KSEQ_DNA_LINE_LENGTH :12-14). The reader's state machine is restated
 * character by character; the stream layer (16 KiB refills, ks_getc / ks_getuntil2, kseq.h:69-150) only
 * moves bytes and is replaced by an index into the buffer. Parity status: PINNED against the reference's own
 * kseq.h compiled into oracle/_ref/libswsem_ref.so (ref_harness.cpp: reff_parse), tests/test_fasta_input.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "oracle.h"

#define LINE_UNKNOWN 0ull                 /* DNA_LINE_LENGTH_UNKNOWN, kseq.h:34 */
#define LINE_BAD UINT64_MAX               /* DNA_NOT_WELLFORMED (-1 in a uint64_t member), kseq.h:35,340 */

/* Parses the whole file. seqOut receives the sequences back to back (capacity >= n). rec[k] = {header
 * offset (byte after '>'), header length, sequence offset in seqOut, sequence length}. Returns the number of
 * records read before the loop ended; *status = 0 (clean EOF), -3 (not FASTA), -4 (lines not well-formed:
 * the record that detects it is NOT counted, as the reference's loop stops on it). */
uint64_t orc_fasta_parse(const uint8_t *f, uint64_t n, int uppercase, uint8_t *seqOut, orc_fasta_record *rec, uint64_t recCap,
                         uint64_t *seqBytes, uint64_t *dnaLineLenOut, int *status) {
    uint64_t pos = 0, out = 0, nrec = 0;
    uint64_t dnaLineLen = LINE_UNKNOWN, maxLast = 0;
    int last_char = 0;
    *status = 0;
    for (;;) {
        /* :237-241 */
        if (last_char == 0) {
            if (pos >= n) break;                                    /* c < 0: end of file */
            int c = f[pos++];
            if (c != '>') { *status = -3; break; }
            last_char = c;
        }
        /* :243 ks_getuntil(KS_SEP_LINE, &name, loosy = false): the rest of the header line; -1 when nothing is left */
        if (pos >= n) break;                                        /* "normal exit: EOF" (r = -1): a lone '>' at the very end */
        uint64_t hs = pos;
        while (pos < n && f[pos] != '\n') pos++;
        uint64_t hlen = pos - hs;
        if (pos < n) pos++;                                         /* consume the newline */
        /* :248-257 */
        const uint64_t seqStart = out;
        uint64_t b = out;
        int c = -1;
        while (pos < n && (c = f[pos++]) != '>') {
            if (c == '\n') { dnaLineLen = LINE_BAD; c = -1; continue; }          /* an empty line */
            if (out > b) {
                const uint64_t d = out - b;
                dnaLineLen = dnaLineLen == LINE_UNKNOWN ? d : (dnaLineLen == d ? d : LINE_BAD);
            }
            b = out;
            seqOut[out++] = (uint8_t) c;
            while (pos < n && f[pos] != '\n') seqOut[out++] = f[pos++];          /* ks_getuntil2(..., append) */
            if (pos < n) pos++;
            c = -1;
        }
        /* :258-264 */
        if (out > b) {
            const uint64_t d = out - b;
            if (d > maxLast) maxLast = d;
        }
        dnaLineLen = (!dnaLineLen || dnaLineLen >= maxLast) ? dnaLineLen : LINE_BAD;
        if (dnaLineLen == LINE_BAD) { *status = -4; out = seqStart; break; }
        last_char = (c == '>') ? c : 0;                              /* :266 (last_char keeps '>' until the next call resets nothing else) */
        if (c != '>') last_char = 0;
        /* processTarget: uppercaseDNA (:361-362) */
        if (uppercase) for (uint64_t i = seqStart; i < out; i++) seqOut[i] = (uint8_t) toupper(seqOut[i]);
        if (nrec < recCap) { rec[nrec].headerOff = hs; rec[nrec].headerLen = hlen; rec[nrec].seqOff = seqStart; rec[nrec].seqLen = out - seqStart; }
        nrec++;
        if (c != '>') {                                              /* the stream is exhausted: the next call returns -1 */
            if (pos >= n) break;
        }
    }
    *seqBytes = out;
    /* KSEQ_DNA_LINE_LENGTH, MGMP.cpp:12-14 */
    *dnaLineLenOut = dnaLineLen == LINE_BAD ? 0 : dnaLineLen;
    return nrec;
}
