/* Test program for the drop-in layer: a miniature of the reference's decode_file()
 * (jpeg2png.c:120-172) that provides the host callbacks the reference program provides
 * (logger_log logger.c:20, progressbar_inc progressbar.c:53) and calls compute() with the
 * reference's own struct layout.  Input: a raw dump written by the test; output: raw planes
 * and the CSV log in the reference's format (logger.c:13,23). */
#define _DEFAULT_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <time.h>

#include "jpeg2png_amd_compute.h"

static unsigned ticks = 0, tick_times = 0;      /* tick_times: ticks that came more than 10 us after the one before */
static double last_tick_us = -1e30;

void progressbar_inc(struct progressbar *pb)
{
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        const double now_us = (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
        if(now_us - last_tick_us > 10.) { tick_times++; }
        last_tick_us = now_us;
        pb->current++;
        ticks++;
}

void logger_log(struct logger *log, double objective, double prob_dist, double tv, double tv2)
{
        if(log->f) {
                fprintf(log->f, "%s,%u,%u,%f,%f,%f,%f\n", log->filename, log->channel, log->iteration, objective, prob_dist, tv, tv2);
        }
}

static void rd(void *p, size_t n, FILE *f)
{
        if(fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char **argv)
{
        if(argc != 4 && argc != 5) { fprintf(stderr, "usage: dropin in.bin out.bin log.csv [tiled:<nband>]\n"); return 2; }
        FILE *in = fopen(argv[1], "rb");
        if(!in) { perror("in"); return 2; }
        unsigned nch, iterations;
        float weight, pweight[3];
        rd(&nch, 4, in); rd(&iterations, 4, in); rd(&weight, 4, in); rd(pweight, 12, in);
        struct coef coefs[3];
        for(unsigned c = 0; c < nch; c++) {
                unsigned g[4];
                rd(g, 16, in);
                coefs[c].w = g[0]; coefs[c].h = g[1]; coefs[c].w_samp = g[2]; coefs[c].h_samp = g[3];
                size_t n = (size_t)g[0] * g[1];
                coefs[c].data = malloc(n * sizeof(int16_t));
                coefs[c].fdata = aligned_alloc(16, n * sizeof(float));       /* alloc_simd */
                rd(coefs[c].data, n * sizeof(int16_t), in);
                rd(coefs[c].fdata, n * sizeof(float), in);
                rd(coefs[c].quant_table, 128, in);
        }
        fclose(in);
        struct logger log;
        log.f = fopen(argv[3], "w");
        log.filename = "dropin";
        log.channel = nch == 3 ? 3 : 0;
        log.iteration = 0;
        fprintf(log.f, "filename,channel,iteration,objective,prob_dist,tv,tv2\n");
        struct progressbar pb = {0, iterations};

        if(argc == 5 && strncmp(argv[4], "tiled:", 6) == 0) {
                /* the multi-GPU entry point with <nband> row bands, all on device 0: the C-level row tiling as far
                 * as a single-GPU box can exercise it */
                extern const char *j2p_last_error(void);
                int devs[32] = {0};
                int rc = j2p_compute_tiled((unsigned)atoi(argv[4] + 6), devs, nch, coefs, &log, &pb, weight, pweight, iterations);
                if(rc != 0) { fprintf(stderr, "jpeg2png: %s\n", j2p_last_error()); return 1; }
        } else {
                compute(nch, coefs, &log, &pb, weight, pweight, iterations);
        }

        fclose(log.f);
        FILE *out = fopen(argv[2], "wb");
        fwrite(&ticks, 4, 1, out);
        fwrite(&tick_times, 4, 1, out);
        for(unsigned c = 0; c < nch; c++) {
                fwrite(&coefs[c].w, 4, 1, out);
                fwrite(&coefs[c].h, 4, 1, out);
                fwrite(coefs[c].fdata, sizeof(float), (size_t)coefs[c].w * coefs[c].h, out);
                free(coefs[c].fdata);          /* free_simd, jpeg2png.c:169 */
                free(coefs[c].data);
        }
        fclose(out);
        return 0;
}
