/*
 * jpeg2png_gpu — command-line driver with the reference's flags and behaviour
 * (jpeg2png.c:177-357, decode_file :120-172) on top of libjpeg2png_amd.so.
 *
 * What stays on the host: option parsing, reading the quantised DCT coefficients with
 * libjpeg (the contract of jpeg.c:22-80), writing the PNG with libpng (png.c:20-78).
 * What moves to the GPU: decode_coefficients + unbox (jpeg.c:83-92, box.c:5), the whole
 * solver, the luma fix-up and the YCbCr->RGB conversion (jpeg2png.c:156-159, png.c:37-62),
 * so that only int16 coefficients go up and only RGB bytes come down.
 *
 * Differences from the reference, on purpose:
 *   -t threads   = number of input files in flight (host threads reading / writing files; each file's GPU work
 *                  is a job of the library's batch engine, j2p_batch_*), instead of an OpenMP thread count;
 *                  default: the number of online cores, as OpenMP's default is (jpeg2png.c:246-257, :330)
 *   J2P_DEVICE / J2P_DEVICES environment: GPU index, or a comma list.  Files are spread over the listed GPUs; with
 *                  fewer files than GPUs the GPUs are shared out among the files and every image is row-tiled
 *                  over its share (one band per GPU; the library uses fewer bands, or one GPU, for images too
 *                  small to pay for the cross-band schedule: at least 2 Mpixel per band) — same pixels either way
 * Messages and exit codes follow the reference ("jpeg2png: <message>", EXIT_FAILURE).
 */
#define _POSIX_C_SOURCE 200809L
#include <getopt.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <jpeglib.h>
#include <png.h>

#include "jpeg2png_amd.h"

#define VERSION_STRING "1.01-gfx950"
#define BAR_WIDTH 70u

/* ---- progress bar (same look as progressbar.c:16-29, 56-66) and die() (utils.c:11-40) ---- */

static pthread_mutex_t ui_lock = PTHREAD_MUTEX_INITIALIZER;
static bool bar_on = false;
static unsigned bar_cur = 0, bar_max = 1;

static void bar_draw(void)
{
        unsigned fill = BAR_WIDTH * bar_cur / bar_max;
        printf("\r[");
        for(unsigned i = 0; i < BAR_WIDTH; i++) { putchar(i < fill ? '#' : ' '); }
        printf("] %3u%%", 100 * bar_cur / bar_max);
        fflush(stdout);
}

static void bar_clear(void)
{
        printf("\r%*s\r", (int)(BAR_WIDTH + 7), "");
        fflush(stdout);
}

static void bar_add(unsigned n)
{
        pthread_mutex_lock(&ui_lock);
        if(bar_on) {
                unsigned of = BAR_WIDTH * bar_cur / bar_max, op = 100 * bar_cur / bar_max;
                bar_cur += n;
                if(of != BAR_WIDTH * bar_cur / bar_max || op != 100 * bar_cur / bar_max) { bar_draw(); }
        }
        pthread_mutex_unlock(&ui_lock);
}

static void die_va(bool with_errno, const char *fmt, va_list l)
{
        pthread_mutex_lock(&ui_lock);
        if(bar_on) { bar_clear(); bar_on = false; }
        fprintf(stderr, "jpeg2png: ");
        vfprintf(stderr, fmt, l);
        if(with_errno) { fprintf(stderr, ": "); perror(NULL); } else { fprintf(stderr, "\n"); }
        exit(EXIT_FAILURE);
}

static void die(const char *fmt, ...)
{
        va_list l;
        va_start(l, fmt);
        die_va(false, fmt, l);
}

static void die_perror(const char *fmt, ...)
{
        va_list l;
        va_start(l, fmt);
        die_va(true, fmt, l);
}

/* ---- JPEG coefficients in (what jpeg.c:22-80 delivers) ---- */

struct component {
        unsigned w, h, w_samp, h_samp;
        int16_t *data;
        uint16_t quant[64];
};

struct jpeg_in {
        unsigned w, h;
        struct component c[3];
};

/* libjpeg's output_message hook: warnings (a truncated file, extraneous bytes ...) are printed and the
 * decode carries on, exactly like die_output_message of jpeg.c:14-19 — which, despite its name, returns.
 * Like die_message_start (utils.c:10-16) it takes the progress bar down for good.  Fatal errors keep
 * libjpeg's default error_exit: output_message, then exit(EXIT_FAILURE). */
static void jpeg_message(j_common_ptr info)
{
        char buf[JMSG_LENGTH_MAX];
        (*info->err->format_message)(info, buf);
        pthread_mutex_lock(&ui_lock);
        if(bar_on) { bar_clear(); bar_on = false; }
        fprintf(stderr, "jpeg2png: libjpeg error: %s\n", buf);
        pthread_mutex_unlock(&ui_lock);
}

static void read_coefficients(FILE *in, struct jpeg_in *jp)
{
        struct jpeg_decompress_struct d;
        struct jpeg_error_mgr err;
        d.err = jpeg_std_error(&err);
        err.output_message = jpeg_message;
        jpeg_create_decompress(&d);
        jpeg_stdio_src(&d, in);
        jpeg_read_header(&d, TRUE);
        jp->w = d.image_width;
        jp->h = d.image_height;
        if(d.num_components != 3) { die("only 3 component jpegs are supported"); }
        for(int c = 0; c < 3; c++) {
                int t = d.comp_info[c].quant_tbl_no;
                if(t < 0 || t >= NUM_QUANT_TBLS) { die("weird jpeg: invalid quant_tbl_no"); }
                JQUANT_TBL *tbl = d.quant_tbl_ptrs[t];
                if(!tbl) { die("weird jpeg: no quant table pointer"); }
                for(int j = 0; j < 64; j++) {
                        if(tbl->quantval[j] == 0) { die("invalid quantization table"); }
                        jp->c[c].quant[j] = tbl->quantval[j];
                }
        }
        jvirt_barray_ptr *arrays = jpeg_read_coefficients(&d);
        for(int c = 0; c < 3; c++) {
                jpeg_component_info *ci = &d.comp_info[c];
                struct component *k = &jp->c[c];
                k->w = ci->width_in_blocks * 8;
                k->h = ci->height_in_blocks * 8;
                k->w_samp = d.max_h_samp_factor / ci->h_samp_factor;
                k->h_samp = d.max_v_samp_factor / ci->v_samp_factor;
                if(k->h / 8 != (jp->h / k->h_samp + 7) / 8) { die("jpeg invalid coef h size"); }
                if(k->w / 8 != (jp->w / k->w_samp + 7) / 8) { die("jpeg invalid coef w size"); }
                if(SIZE_MAX / k->h / k->w / k->h_samp / k->w_samp < 6) { die("jpeg is too big to fit in memory"); }
                k->data = malloc(sizeof(int16_t) * (size_t)k->w * k->h);
                if(!k->data) { die("could not allocate memory for coefs"); }
                int16_t *dst = k->data;
                for(unsigned by = 0; by < ci->height_in_blocks; by++) {
                        JBLOCKARRAY row = d.mem->access_virt_barray((j_common_ptr)&d, arrays[c], by, 1, FALSE);
                        memcpy(dst, row[0][0], sizeof(int16_t) * 64 * ci->width_in_blocks);
                        dst += 64 * ci->width_in_blocks;
                }
        }
        jpeg_destroy_decompress(&d);
}

/* ---- PNG out (the file side of png.c:20-78; pixels arrive already converted) ---- */

static void png_fatal(png_structp png, png_const_charp msg)
{
        (void)png;
        die("%s", msg);
}

static void write_rgb_png(FILE *out, unsigned w, unsigned h, unsigned bits, uint8_t *pixels)
{
        png_structp png = png_create_write_struct(PNG_LIBPNG_VER_STRING, NULL, png_fatal, NULL);
        if(!png) { die("could not initialize PNG write struct"); }
        png_infop info = png_create_info_struct(png);
        if(!info) { die("could not initialize PNG info struct"); }
        png_init_io(png, out);
        png_set_IHDR(png, info, w, h, (int)bits, PNG_COLOR_TYPE_RGB, PNG_INTERLACE_NONE, PNG_COMPRESSION_TYPE_BASE,
                     PNG_FILTER_TYPE_BASE);
        png_write_info(png, info);
        png_bytep *rows = malloc(sizeof(*rows) * h);
        if(!rows) { die("allocation failure"); }
        size_t stride = (size_t)w * 3 * (bits / 8);
        for(unsigned y = 0; y < h; y++) { rows[y] = pixels + y * stride; }
        png_write_image(png, rows);
        free(rows);
        png_write_end(png, info);
        png_destroy_write_struct(&png, &info);
}

/* ---- one file: decode_file (jpeg2png.c:120-172) ---- */

struct options {
        unsigned iterations[3];
        float weights[3], pweights[3];
        unsigned png_bits;
        bool joint, quiet;
        bool tile;              /* fewer files than GPUs: every image row-tiled over its share of them */
        unsigned nfiles;
        FILE *csv;
        int ndev, devs[16];
};

static pthread_mutex_t csv_lock = PTHREAD_MUTEX_INITIALIZER;
/* the library's batch engine, created when the first file has been read (a file that cannot be read fails on its
 * own message, with or without a GPU) */
static j2p_batch *batch = NULL;
static pthread_once_t batch_once = PTHREAD_ONCE_INIT;
static unsigned batch_ndev = 1, batch_slots = 1;
static int batch_devs[16];
static int batch_rc = J2P_OK;
static char batch_err[512];

static void batch_start(void)
{
        batch_rc = j2p_batch_create(&batch, batch_ndev, batch_devs, batch_slots);
        if(batch_rc != J2P_OK) { snprintf(batch_err, sizeof(batch_err), "%s", j2p_last_error()); }
}

static void gpu_check(int rc)
{
        if(rc != J2P_OK) { die("%s", j2p_last_error()); }
}

/* callbacks of a batch job (called on the library's worker thread): CSV rows (logger.c:20-28) and progress ticks */
struct job_ctx {
        const struct options *o;
        const char *name;
};

static void job_rows(void *user, unsigned channel, unsigned first, unsigned n, const j2p_log_row *rows)
{
        struct job_ctx *ctx = user;
        pthread_mutex_lock(&csv_lock);                                 /* critical(write_log), logger.c:22 */
        for(unsigned i = 0; i < n; i++) {
                if(fprintf(ctx->o->csv, "%s,%u,%u,%f,%f,%f,%f\n", ctx->name, channel, first + i, rows[i].objective,
                           rows[i].prob_dist, rows[i].tv, rows[i].tv2) < 0) {
                        die_perror("could not write to csv log");
                }
        }
        pthread_mutex_unlock(&csv_lock);
}

static void job_progress(void *user, unsigned n)
{
        (void)user;
        bar_add(n);
}

/* decode_file (jpeg2png.c:120-172): the coefficients go to the library's batch engine, which decodes, solves
 * (one joint compute(3, ...) or three compute(1, ...), jpeg2png.c:141-152) and converts on a GPU slot of its own
 * while this thread's neighbours read their JPEGs and deflate their PNGs */
static void decode_file(const char *infile, const char *outfile, const struct options *o, unsigned index)
{
        FILE *in = fopen(infile, "rb");
        if(!in) { die_perror("could not open input file `%s`", infile); }
        struct jpeg_in jp;
        read_coefficients(in, &jp);
        fclose(in);

        struct job_ctx ctx = {o, infile};
        j2p_job job;
        memset(&job, 0, sizeof(job));
        job.nchannel = 3;
        for(int c = 0; c < 3; c++) {
                job.planes[c].w = jp.c[c].w;
                job.planes[c].h = jp.c[c].h;
                job.planes[c].w_samp = jp.c[c].w_samp;
                job.planes[c].h_samp = jp.c[c].h_samp;
                job.planes[c].data = jp.c[c].data;
                job.planes[c].fdata = NULL;                             /* decoded on the device */
                job.planes[c].quant_table = jp.c[c].quant;
                job.weight[c] = o->weights[c];
                job.pweight[c] = o->pweights[c];
                job.iterations[c] = o->iterations[c];
        }
        job.separate = !o->joint;
        job.tile = o->tile;
        {
                /* (tests row-tile small images: the pixel gate of the library can be lowered from outside the program) */
                const char *gate = getenv("J2P_TILE_MIN_BAND_PIXELS");
                if(gate && *gate) {
                        const unsigned long long v = strtoull(gate, NULL, 10);
                        job.tile_min_band_pixels = v ? (size_t)v : (size_t)-1;
                }
        }
        if(o->tile && o->nfiles > 1) {
                /* several files, still fewer than GPUs: file i gets GPUs [i n / f, (i + 1) n / f) of the list */
                job.tile_first = index * (unsigned)o->ndev / o->nfiles;
                job.tile_count = (index + 1) * (unsigned)o->ndev / o->nfiles - job.tile_first;
        }
        job.out_bits = o->png_bits;
        job.out_w = jp.w;
        job.out_h = jp.h;
        size_t bytes = (size_t)jp.w * jp.h * 3 * (o->png_bits / 8);
        uint8_t *pixels = malloc(bytes);
        if(!pixels) { die("could not allocate image data"); }
        job.out_rgb = pixels;
        job.on_rows = o->csv ? job_rows : NULL;
        job.on_progress = o->quiet ? NULL : job_progress;
        job.user = &ctx;
        int ticket = 0;
        pthread_once(&batch_once, batch_start);
        if(batch_rc != J2P_OK) { die("%s", batch_err); }
        gpu_check(j2p_batch_submit(batch, &job, &ticket));
        gpu_check(j2p_batch_wait(batch, ticket));
        for(int c = 0; c < 3; c++) { free(jp.c[c].data); }
        FILE *out = fopen(outfile, "wb");
        if(!out) { die_perror("could not open output file `%s`", outfile); }
        write_rgb_png(out, jp.w, jp.h, o->png_bits, pixels);
        fclose(out);
        free(pixels);
}

/* ---- file-level parallelism (jpeg2png.c:330: omp parallel for over files) ---- */

struct work {
        unsigned nin, next;
        char **in, **out;
        const struct options *o;
        pthread_mutex_t lock;
};

static void *worker(void *arg)
{
        struct work *w = arg;
        for(;;) {
                pthread_mutex_lock(&w->lock);
                unsigned i = w->next++;
                pthread_mutex_unlock(&w->lock);
                if(i >= w->nin) { break; }
                decode_file(w->in[i], w->out[i], w->o, i);
        }
        return NULL;
}

static void usage(void)
{
        printf("usage: jpeg2png_gpu picture.jpg ... [-o picture.png] ... [flags...]\n\n"
               "  -o, --output FILE            output file (overwritten); zero times or once per input\n"
               "                               default: input name with the extension .png\n"
               "  -f, --force                  overwrite default-named outputs too\n"
               "  -w, --second-order-weight W[,Wcb,Wcr]\n"
               "                               TGV weight, 0 = plain total variation (default 0.3; chroma 0\n"
               "                               with -s); three values need -s\n"
               "  -p, --probability-weight P[,Pcb,Pcr]\n"
               "                               DCT-coefficient distance weight (default 0.001)\n"
               "  -i, --iterations N[,Ncb,Ncr] optimisation steps (default 50); three values need -s\n"
               "  -s, --separate-components    optimise Y, Cb, Cr independently (three GPU streams)\n"
               "  -t, --threads N              input files processed concurrently (default: online cores)\n"
               "  -1, --16-bits-png            16-bit PNG\n"
               "  -c, --csv-log FILE           per-iteration objective log\n"
               "  -q, --quiet                  no progress bar\n"
               "  -h, --help    -V, --version\n"
               "environment: J2P_DEVICE=n or J2P_DEVICES=a,b,... selects the GPU(s); fewer files than GPUs:\n"
               "             large images are row-tiled over their share of them\n");
        exit(EXIT_FAILURE);
}

int main(int argc, char **argv)
{
        static const struct option longopts[] = {
                {"help", no_argument, NULL, 'h'}, {"version", no_argument, NULL, 'V'}, {"output", required_argument, NULL, 'o'},
                {"force", no_argument, NULL, 'f'}, {"csv-log", required_argument, NULL, 'c'},
                {"threads", required_argument, NULL, 't'}, {"quiet", no_argument, NULL, 'q'},
                {"separate-components", no_argument, NULL, 's'}, {"16-bits-png", no_argument, NULL, '1'},
                {"iterations", required_argument, NULL, 'i'}, {"probability-weight", required_argument, NULL, 'p'},
                {"second-order-weight", required_argument, NULL, 'w'}, {NULL, 0, NULL, 0}};
        struct options o = {.iterations = {50, 50, 50}, .weights = {0.3f, 0.f, 0.f}, .pweights = {0.001f, 0.001f, 0.001f},
                            .png_bits = 8, .joint = true, .quiet = false, .tile = false, .csv = NULL, .ndev = 1, .devs = {0}};
        const char *w_arg = NULL, *p_arg = NULL, *i_arg = NULL, *t_arg = NULL, *c_arg = NULL;
        char **outs = calloc((size_t)argc, sizeof(*outs));
        unsigned nout = 0;
        bool force = false, help = false, version = false;
        int ch;
        while((ch = getopt_long(argc, argv, "h?Vo:fc:t:qs1i:p:w:", longopts, NULL)) != -1) {
                switch(ch) {
                case 'V': version = true; break;
                case 'o': outs[nout++] = optarg; break;
                case 'f': force = true; break;
                case 'c': c_arg = optarg; break;
                case 't': t_arg = optarg; break;
                case 'q': o.quiet = true; break;
                case 's': o.joint = false; break;
                case '1': o.png_bits = 16; break;
                case 'i': i_arg = optarg; break;
                case 'p': p_arg = optarg; break;
                case 'w': w_arg = optarg; break;
                default: help = true; break;
                }
        }
        if(version) {
                printf("jpeg2png_gpu version " VERSION_STRING " (%s)\n", j2p_version());
                exit(EXIT_FAILURE);                                     /* sic: jpeg2png.c:195-198 */
        }
        unsigned nin = (unsigned)(argc - optind);
        if(nin == 0 || help) { usage(); }
        char **ins = argv + optind;

        if(w_arg) {
                int n = sscanf(w_arg, "%f,%f,%f", &o.weights[0], &o.weights[1], &o.weights[2]);
                if(n == 3) { if(o.joint) { die("different weights are only possible when using separated components"); } }
                else if(n != 1) { die("invalid weight"); }
        }
        if(p_arg) {
                int n = sscanf(p_arg, "%f,%f,%f", &o.pweights[0], &o.pweights[1], &o.pweights[2]);
                if(n == 1) { o.pweights[1] = o.pweights[2] = o.pweights[0]; }
                else if(n != 3) { die("invalid probability weight"); }
        }
        if(i_arg) {
                int n = sscanf(i_arg, "%u,%u,%u", &o.iterations[0], &o.iterations[1], &o.iterations[2]);
                if(n == 3) { if(o.joint) { die("different iteration counts are only possible when using separated components"); } }
                else if(n == 1) { o.iterations[1] = o.iterations[2] = o.iterations[0]; }
                else { die("invalid number of iterations"); }
        }
        /* the reference leaves the thread count to OpenMP, i.e. one per online core (jpeg2png.c:246-257) */
        long cores = sysconf(_SC_NPROCESSORS_ONLN);
        unsigned threads = cores > 0 ? (unsigned)cores : 1u;
        if(t_arg) {
                if(sscanf(t_arg, "%u", &threads) != 1 || threads == 0) { die("invalid number of threads"); }
        }
        if(c_arg) {
                o.csv = fopen(c_arg, "wb");
                if(!o.csv) { die_perror("could not open csv log `%s`", c_arg); }
                if(fprintf(o.csv, "filename,channel,iteration,objective,prob_dist,tv,tv2\n") < 0) {
                        die_perror("could not write to csv log");
                }
        }
        const char *devs = getenv("J2P_DEVICES");
        if(!devs) { devs = getenv("J2P_DEVICE"); }
        if(devs && *devs) {
                o.ndev = 0;
                char *copy = strdup(devs), *save = NULL;
                for(char *tok = strtok_r(copy, ",", &save); tok && o.ndev < 16; tok = strtok_r(NULL, ",", &save)) {
                        o.devs[o.ndev++] = atoi(tok);
                }
                free(copy);
                if(o.ndev == 0) { o.ndev = 1; o.devs[0] = 0; }
        }

        if(!(nout == 0 || nout == nin)) { die("must give output file names for all input files or none"); }
        char **outfiles = malloc(sizeof(*outfiles) * nin);
        if(!outfiles) { die("could not allocate outfiles"); }
        for(unsigned i = 0; i < nin; i++) {
                if(nout) { outfiles[i] = outs[i]; continue; }
                const char *infile = ins[i];
                FILE *in = fopen(infile, "rb");
                if(!in) { die("could not open input file `%s`", infile); }
                fclose(in);
                size_t l = strlen(infile), e = l;                       /* jpeg2png.c:291-301 */
                if(l >= 5 && memcmp(".jpeg", infile + l - 5, 5) == 0) { e = l - 5; }
                else if(l >= 4 && memcmp(".jpg", infile + l - 4, 4) == 0) { e = l - 4; }
                char *outfile = malloc(e + 5);
                if(!outfile) { die("could not allocate outfile"); }
                memcpy(outfile, infile, e);
                memcpy(outfile + e, ".png", 5);
                if(!force) {
                        FILE *exists = fopen(outfile, "rb");
                        if(exists) { die("not overwriting output file `%s`", outfile); }
                }
                FILE *probe = fopen(outfile, "wb");
                if(!probe) { die("could not open output file `%s`", outfile); }
                fclose(probe);
                remove(outfile);
                outfiles[i] = outfile;
        }

        if(!o.quiet) {
                pthread_mutex_lock(&ui_lock);
                bar_max = o.joint ? nin * o.iterations[0] : nin * (o.iterations[0] + o.iterations[1] + o.iterations[2]);
                if(bar_max == 0) { bar_max = 1; }
                bar_cur = 0;
                bar_on = true;
                bar_draw();
                pthread_mutex_unlock(&ui_lock);
        }

        struct work w = {.nin = nin, .next = 0, .in = ins, .out = outfiles, .o = &o};
        pthread_mutex_init(&w.lock, NULL);
        if(threads > nin) { threads = nin; }
        /* as many GPU slots as files in flight, spread over the devices of J2P_DEVICES; with fewer files than GPUs
         * the GPUs are shared out among the files and the library tiles every image over its share (decode_file ->
         * compute of one large image, jpeg2png.c:141-152) when the image is large enough for that to pay */
        o.tile = o.ndev > 1 && nin < (unsigned)o.ndev;
        o.nfiles = nin;
        batch_ndev = (unsigned)o.ndev;
        memcpy(batch_devs, o.devs, sizeof(batch_devs));
        batch_slots = (threads + batch_ndev - 1) / batch_ndev;
        if(batch_slots > 16) { batch_slots = 16; }
        pthread_t *tid = malloc(sizeof(*tid) * threads);
        for(unsigned t = 1; t < threads; t++) { pthread_create(&tid[t], NULL, worker, &w); }
        worker(&w);
        for(unsigned t = 1; t < threads; t++) { pthread_join(tid[t], NULL); }
        free(tid);
        if(batch) { j2p_batch_destroy(batch); }

        if(!nout) { for(unsigned i = 0; i < nin; i++) { free(outfiles[i]); } }
        free(outfiles);
        free(outs);
        if(!o.quiet) {
                pthread_mutex_lock(&ui_lock);
                bar_clear();
                bar_on = false;
                pthread_mutex_unlock(&ui_lock);
        }
        if(o.csv) { fclose(o.csv); }
        return 0;
}
