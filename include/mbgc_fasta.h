/* mbgc_fasta.h — C ABI of the MI355X input stage of `mbgc c`: what the reference does per
 * target file between "the file's bytes are in memory" and "its contigs are char arrays for matchTexts":
 *
 *   kseq_init + while (KSEQ_READ(seq) >= 0) { readHeader; seq->seq.s / seq->seq.l ... }   matching/MultipleGenomeMatchingProcessor.cpp:349-372
 *   KSEQ_READ = kseq_read_lossless_fasta                                                    :9-10, utils/kseq.h:233-274
 *   validate_kseq_status (-3 "expected FASTA format", -4 "inconsistent line length")         :16-35
 *   KSEQ_DNA_LINE_LENGTH                                                                     :12-14
 *   params->uppercaseDNA -> PgHelpers::upperSequence                                         :361-362, utils/helper.cpp:447-453
 *
 * The files of a round sit back to back in one device buffer (whole-file reads / inflated .gz land in pinned
 * memory and are copied up by the caller). One call strips the headers and the newlines of all of them and
 * leaves every file's contigs back to back in HBM — exactly the layout swsem_match_batch_dev takes — plus the
 * record table (header bytes, contig offsets), the detected DNA line length and the kseq status per file.
 * gzip inflate (input_with_libdeflate_wrapper.cpp) stays on the host. No CPU fallback. */
#ifndef MBGC_FASTA_H
#define MBGC_FASTA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbgc_fasta mbgc_fasta_t;

typedef struct {
    uint64_t headerOff, headerLen;   /* the header line without '>' and '\n' (seq->name), bytes of the FILE */
    uint64_t seqOff, seqLen;         /* the contig (seq->seq), bytes of the file's part of the output       */
} mbgc_fasta_record_t;

#define MBGC_FASTA_OK 0
#define MBGC_FASTA_ENOTFASTA (-3)    /* kseq status -3: the file does not start with '>'                     */
#define MBGC_FASTA_ELINES (-4)       /* kseq status -4: empty line / inconsistent line lengths (kseq.h:251-265) */

int mbgc_fasta_create(mbgc_fasta_t **out, int device);
void mbgc_fasta_destroy(mbgc_fasta_t *p);
const char *mbgc_fasta_last_error(void);

/* files_dev[fileOff[f] .. fileOff[f+1]) = file f (host array of nf + 1 offsets). seq_out_dev (capacity outCap
 * bytes; fileOff[nf] - fileOff[0] always suffices) receives the contigs; file f's start at seqBase[f] and
 * seqBase[nf] is the total. records (capacity recCap; at most one per two input bytes) receives the records of
 * all files in order, file f's at [recBase[f], recBase[f+1]). status[f] is the kseq status the reference's read
 * loop ends with (0 = clean end of file); for a file with status != 0 — the reference prints its message and
 * exits there — records and sequence bytes of that file are unspecified. dnaLineLen[f] = KSEQ_DNA_LINE_LENGTH.
 * Returns 0, or a negative error of its own (capacity, HIP) with mbgc_fasta_last_error(); when the record table
 * is too small (-104) recBase[nf] holds the number of entries needed. Synchronous. */
int mbgc_fasta_parse_batch_dev(mbgc_fasta_t *p, const uint8_t *files_dev, const uint64_t *fileOff, int nf, int uppercaseDNA,
                               uint8_t *seq_out_dev, uint64_t outCap, uint64_t *seqBase,
                               mbgc_fasta_record_t *records, uint64_t recCap, uint64_t *recBase,
                               uint64_t *dnaLineLen, int *status);

/* One file that sits in host memory and whose contigs the host needs as bytes — the first file of the list, whose
 * sequences become the initial reference and the head of the literal stream (loadG0Ref, MGMP.cpp:66-150): uploaded,
 * parsed by the same kernels, downloaded. seq_out_host has capacity n; records/recCap as above; *nrec, *seqBytes,
 * *dnaLineLen, *status as the batch call reports them for its single file. */
int mbgc_fasta_parse_host(mbgc_fasta_t *p, const uint8_t *file_host, uint64_t n, int uppercaseDNA, uint8_t *seq_out_host,
                          uint64_t *seqBytes, mbgc_fasta_record_t *records, uint64_t recCap, uint64_t *nrec,
                          uint64_t *dnaLineLen, int *status);

/* The way into HBM ("fed by pinned-memory reads"): page-locked host memory for the reader threads to
 * read files into, and its copy to the device on the input stage's own stream — it returns when the bytes have arrived
 * and waits for nothing the matcher has queued. */
int mbgc_fasta_host_alloc(mbgc_fasta_t *p, uint64_t bytes, void **out);
int mbgc_fasta_host_free(mbgc_fasta_t *p, void *ptr);
int mbgc_fasta_upload(mbgc_fasta_t *p, uint8_t *dst_dev, const void *src_host, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
