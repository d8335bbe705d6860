// fsgpu-modules: command-line front end for the three hot-path modules (same sub-command names as the reference binary)
#include "fshost.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, const char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <search|prefilter|ungappedprefilter|structurealign|structurerescorediagonal|makepaddedseqdb|gpuserver|convertalis|indexdb|createindex> <args...>\n", argv[0]);
        return EXIT_FAILURE;
    }
    if (!strcmp(argv[1], "ungappedprefilter")) return fsmod_ungappedprefilter(argc - 2, argv + 2);
    if (!strcmp(argv[1], "prefilter")) return fsmod_prefilter(argc - 2, argv + 2);
    if (!strcmp(argv[1], "search")) return fsmod_search(argc - 2, argv + 2);
    if (!strcmp(argv[1], "structurealign")) return fsmod_structurealign(argc - 2, argv + 2);
    if (!strcmp(argv[1], "gpuserver")) return fsmod_gpuserver(argc - 2, argv + 2);
    if (!strcmp(argv[1], "structurerescorediagonal") || !strcmp(argv[1], "structureungappedalign")) return fsmod_structurerescorediagonal(argc - 2, argv + 2);
    if (!strcmp(argv[1], "makepaddedseqdb")) return fsmod_makepaddedseqdb(argc - 2, argv + 2);
    if (!strcmp(argv[1], "convertalis")) return fsmod_convertalis(argc - 2, argv + 2);
    if (!strcmp(argv[1], "indexdb")) return fsmod_indexdb(argc - 2, argv + 2);
    if (!strcmp(argv[1], "createindex")) return fsmod_createindex(argc - 2, argv + 2);
    fprintf(stderr, "unknown module %s\n", argv[1]);
    return EXIT_FAILURE;
}
