/* mbgc_copmem.h — C ABI of the `-m3` reverse-complement pass over the literal stream on an MI355X.
 *
 * What the reference runs at the end of `mbgc c -m3`, between the match-finding path and the entropy backend:
 *
 *   MBGC_Encoder::prepareAndCompressStreams            mbgccoder/MBGC_Encoder.cpp:636-638
 *     SimpleSequenceMatcher::rcMatchSequence(targetLiterals[0], rcMapOff, rcMapLen, rcMatchMinLength = 55)
 *                                                      matching/SimpleSequenceMatcher.cpp:165-176
 *       CopMEMMatcher(seq, n, L, minLen)               matching/copmem/CopMEMMatcher.cpp:497-517 (+ :68-144 parameters,
 *                                                      :146-225 bucketed index of every k1-th position)
 *       matchTexts(reverseComplement(seq), destIsSrc, revComplMatching, minLen)   :349-495, :519-540
 *       markAndRemoveExactMatches' post-processing     SimpleSequenceMatcher.cpp:59-62, :91-147, :150-163
 *
 * The index build, the reverse complement and the query scan run on the device (mbgc_amd/csrc/copmem.hip); sorting the
 * few matches and cutting them out of the sequence stays on the host, where the reference does it and where the backend
 * consumes the stream. Results are those of the reference with ONE thread (with several its index keeps whichever 13
 * positions of a crowded bucket its threads deliver first). No CPU fallback. */
#ifndef MBGC_COPMEM_H
#define MBGC_COPMEM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mbgc_copmem mbgc_copmem_t;
typedef struct { uint64_t posSrcText, length, posDestText; } mbgc_copmem_match_t;   /* TextMatch, matching/TextMatchers.h:9-16 */

int mbgc_copmem_create(mbgc_copmem_t **out, int device);
void mbgc_copmem_destroy(mbgc_copmem_t *p);
const char *mbgc_copmem_last_error(void);

/* The matches CopMEMMatcher::matchTexts pushes for the reverse-complemented sequence against the sequence itself, in
 * push order (dest positions in the reverse-complemented text, as the reference holds them before
 * correctDestPositionDueToRevComplMatching). seq_host: the literal stream in host memory. *matches points into a
 * handle-owned buffer valid until the next call. params (may be NULL) receives K, k1, k2, log2(hash size).
 * minMatchLength = UINT32_MAX means "the target length" (SimpleSequenceMatcher.cpp:80-81) and is what MBGC passes; a value below
 * L is refused (-3): which of the shorter matches the reference then reports is decided by its 4-byte pre-filter, which the
 * device path does not model (it is result-neutral for matches of at least the target length). Returns 0, -3 where the
 * reference prints a message and exits (minimal length < 24 or < K, L/K mismatch), or a negative error of its own. */
int mbgc_copmem_rc_matches(mbgc_copmem_t *p, const uint8_t *seq_host, uint64_t n, uint32_t targetMatchLength,
                           uint32_t minMatchLength, const mbgc_copmem_match_t **matches, uint64_t *count, int params[4]);

/* SimpleSequenceMatcher::rcMatchSequence: seq_host is rewritten in place (matched parts replaced by RC_MATCH_MARK),
 * *newLen is its new length; *mapOff / *mapLen point into handle-owned buffers (valid until the next call) holding the
 * rcMapOff / rcMapLen streams. stats (may be NULL): unique matches, matched characters, characters in overlaps. */
int mbgc_copmem_rc_match_sequence(mbgc_copmem_t *p, uint8_t *seq_host, uint64_t n, uint32_t targetMatchLength,
                                  uint32_t minMatchLength, uint64_t *newLen, const uint8_t **mapOff, uint64_t *mapOffLen,
                                  const uint8_t **mapLen, uint64_t *mapLenLen, uint64_t stats[3]);

#ifdef __cplusplus
}
#endif
#endif
