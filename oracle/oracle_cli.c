/* oracle_cli.c -- TEST INFRASTRUCTURE: command-line front end of the CPU
 * oracle (grlbwt_oracle.c).  Usage: oracle_cli INPUT OUTPUT [-a N] [-q]
 * Used by bench.py's cpu_baseline leg ("kind":"port") and by tests.       */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "grlbwt_oracle.h"

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s INPUT OUTPUT [-a 1|2|4|8] [-q]\n", argv[0]); return 2; }
    int w = 1, quiet = 0;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "-a") && i + 1 < argc) w = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-q")) quiet = 1;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char *buf = malloc(sz > 0 ? sz : 1);
    if (fread(buf, 1, sz, f) != (size_t)sz) { perror("read"); return 2; }
    fclose(f);
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    oracle_result *R = oracle_run(buf, (uint64_t)sz / w, w, 0);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (oracle_status(R) == ORACLE_ERR_ILLFORMED) { printf("Error: the file is ill formed\n"); return 1; }
    if (oracle_status(R) != ORACLE_OK) { fprintf(stderr, "oracle error %d\n", oracle_status(R)); return 2; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    fwrite(oracle_out_bytes(R), 1, oracle_out_size(R), o); fclose(o);
    double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    if (!quiet) printf("oracle: %ld bytes, %d rounds, %.3f s, %.3f MB/s\n", sz, oracle_n_rounds(R), s, sz / 1e6 / s);
    oracle_free(R); free(buf);
    return 0;
}
