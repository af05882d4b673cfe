// mbgc-hip: the compress hot path of `mbgc c` (file list -> match/literal byte streams) on one MI355X,
// driven by the C++ host classes that keep the reference's names. The streams are written raw, one file
// per stream, exactly what the reference's developer build dumps with `mbgc-dev v -D` (main.cpp:575-577;
// stream order of MBGC_Decoder.cpp:1085-1112): literals(13) locksPos(14) gapDelta(15) flags(16)
// mapOff(17) mapLen(18) refExtSize(19). Entropy coding (PPMd/LZMA) is the unchanged host backend of the
// reference and is not part of this tool.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "mgmp_driver.h"

static void dump(const std::string &prefix, const char *name, const std::string &data) {
    std::ofstream f(prefix + "." + name, std::ios::binary | std::ios::trunc);
    f.write(data.data(), (std::streamsize) data.size());
}

int main(int argc, char **argv) {
    MBGC_Params params;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "c") continue;
        if (a == "-t1") params.sequentialMatching = true;                        // like `mbgc c -t1` (MBGC_Params.h:867-869)
        else if (a == "-m" && i + 1 < argc) params.setCompressionMode(atoi(argv[++i]));
        else if (a == "-R" && i + 1 < argc) params.roundSize = atoi(argv[++i]);
        else if (a == "-d" && i + 1 < argc) params.device = atoi(argv[++i]);
        else if (a == "-L") params.lazyDecompressionSupport = false;             // disable lazy decompression support
        else if (a == "-U") params.uppercaseDNA = true;                          // MBGC_Params.h: converts bases to uppercase
        else if (a == "--bench") params.benchMode = true;                        // rounds timed with every contig resident in HBM (no streams written)
        else if (a == "--warmup" && i + 1 < argc) params.benchWarmup = atoi(argv[++i]);
        else pos.push_back(a);
    }
    if (pos.size() != 2) {
        fprintf(stderr, "usage: mbgc-hip c [-t1] [-m mode] [-R targetsPerRound] [-d device] [-U] [--bench [--warmup rounds]] <sequencesListFile> <outputPrefix>\n");
        return EXIT_FAILURE;
    }
    std::vector<std::string> files;
    {
        std::ifstream lst(pos[0]);
        if (!lst) { fprintf(stderr, "cannot open sequences list file %s\n", pos[0].c_str()); return EXIT_FAILURE; }
        std::string line;
        while (std::getline(lst, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (!line.empty()) files.push_back(line);
        }
    }
    MBGC_Encoder enc(&params);
    enc.encode(files);
    if (params.benchMode) {
        // the C++ host's own measurement of the hot path (BASELINE.json metric): inputs resident in HBM, rounds of -R targets
        printf("{\"metric\": \"input Gbases/s (compress hot path, C++ host)\", \"value\": %.4f, \"unit\": \"Gbases/s\", \"rounds\": %d, "
               "\"warmup_rounds\": %d, \"targets_per_round\": %d, \"bases\": %llu, \"seconds\": %.6f, \"ms_per_round\": %.4f}\n",
               params.benchBases / params.benchSeconds / 1e9, params.benchRounds, params.benchWarmup, params.roundSize,
               (unsigned long long) params.benchBases, params.benchSeconds, params.benchSeconds * 1e3 / std::max(1, params.benchRounds));
        return 0;
    }
    dump(pos[1], "literals", enc.literals);
    if (params.rcRedundancyRemoval) { dump(pos[1], "rcMapOff", enc.rcMapOff); dump(pos[1], "rcMapLen", enc.rcMapLen); }
    dump(pos[1], "locksPos", enc.locksPosStream);
    dump(pos[1], "gapDelta", enc.gapDeltas);
    dump(pos[1], "flags", enc.gapMismatchesFlags);
    dump(pos[1], "mapOff", enc.mapOff);
    dump(pos[1], "mapOff5th", enc.mapOff5thByte);
    dump(pos[1], "mapLen", enc.mapLen);
    dump(pos[1], "refExtSize", enc.refExtSizeStream);
    printf("exact matches total: %zu\n", enc.exactMatches());
    printf("removed matches breaking gaps total: %zu\n", enc.removedGapBreakingMatchesAll);
    printf("swsMEM unmatched chars: %zu\n", enc.unmatchedChars());
    printf("extensions matched chars: %zu\n", enc.extensionsMatchedCharsAll);
    printf("final unmatched chars: %zu\n", enc.unmatchedChars() - enc.extensionsMatchedCharsAll);
    return 0;
}
