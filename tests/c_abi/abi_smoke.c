/*
 * abi_smoke.c -- the drop-in boundary used from plain C (tests/test_c_abi.py).
 *
 * Compiled with `gcc -std=c99 -Wall -Wextra -Werror -pedantic` against include/llama2_hip.h ONLY
 * (not the test header): proof that the header is valid C and that a C host can drive the whole
 * hot path through it -- the Zig `extern fn` block of INTEGRATION.md binds exactly these symbols.
 *
 *   abi_smoke layout
 *       prints sizeof(l2z_config) and the offset of every field: must equal the reference's
 *       ConfigReader (src/main.zig:17-25: extern struct of 7 x i32 = 28 bytes, no padding).
 *   abi_smoke run <checkpoint.bin> <n_steps> [prompt tokens...]
 *       the reference's main() at temperature 0 in C: read the file like src/main.zig:936-967,
 *       Weights.init / RunState.init, then the loop of :995-1036 -- transformer(token, pos),
 *       next = prompt[pos] or argmax(logits) -- printing one token id per line.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "llama2_hip.h"

static int fail(const char *what)
{
    fprintf(stderr, "abi_smoke: %s: %s\n", what, l2z_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && strcmp(argv[1], "layout") == 0) {
        printf("sizeof %lu\n", (unsigned long)sizeof(l2z_config));
        printf("dim %lu\n", (unsigned long)offsetof(l2z_config, dim));
        printf("hidden_dim %lu\n", (unsigned long)offsetof(l2z_config, hidden_dim));
        printf("n_layers %lu\n", (unsigned long)offsetof(l2z_config, n_layers));
        printf("n_heads %lu\n", (unsigned long)offsetof(l2z_config, n_heads));
        printf("n_kv_heads %lu\n", (unsigned long)offsetof(l2z_config, n_kv_heads));
        printf("vocab_size %lu\n", (unsigned long)offsetof(l2z_config, vocab_size));
        printf("seq_len %lu\n", (unsigned long)offsetof(l2z_config, seq_len));
        printf("abi %d\n", L2Z_ABI_VERSION);
        return 0;
    }
    if (argc < 4 || strcmp(argv[1], "run") != 0) {
        fprintf(stderr, "usage: abi_smoke layout | abi_smoke run <checkpoint.bin> <n_steps> [prompt...]\n");
        return 2;
    }
    {
        FILE *f = fopen(argv[2], "rb");
        int32_t hdr[7];
        l2z_config cfg;
        int shared, n_steps = atoi(argv[3]), n_prompt = argc - 4;
        long bytes;
        size_t n_floats;
        float *data;
        l2z_weights *w = NULL;
        l2z_runstate *s = NULL;
        int token = 1, pos; /* BOS, main.zig:988 */

        if (!f) { perror(argv[2]); return 1; }
        if (fread(hdr, sizeof hdr, 1, f) != 1) { fprintf(stderr, "short header\n"); return 1; }
        shared = hdr[5] > 0;                       /* main.zig:943 */
        memcpy(&cfg, hdr, sizeof cfg);             /* same 28 bytes */
        if (cfg.vocab_size < 0) cfg.vocab_size = -cfg.vocab_size; /* :944 */
        fseek(f, 0, SEEK_END);
        bytes = ftell(f) - (long)sizeof hdr;
        fseek(f, (long)sizeof hdr, SEEK_SET);
        n_floats = (size_t)bytes / sizeof(float);
        data = (float *)malloc(n_floats * sizeof(float));
        if (!data || fread(data, sizeof(float), n_floats, f) != n_floats) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);

        if (l2z_abi_version() != L2Z_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
        if (l2z_weights_init(&cfg, data, n_floats, shared, NULL, &w) != L2Z_OK) return fail("l2z_weights_init");
        free(data); /* the host blob may go as soon as Weights.init returns */
        if (l2z_runstate_init(&cfg, NULL, &s) != L2Z_OK) return fail("l2z_runstate_init");
        if (n_steps > cfg.seq_len) n_steps = cfg.seq_len;
        for (pos = 0; pos < n_steps; pos++) {      /* main.zig:995 */
            int next;
            if (l2z_transformer(token, pos, &cfg, s, w) != L2Z_OK) return fail("l2z_transformer"); /* :996 */
            if (pos < n_prompt) {
                next = atoi(argv[4 + pos]);        /* :999-1000 */
            } else if (l2z_argmax(s, &next) != L2Z_OK) {  /* :1003 */
                return fail("l2z_argmax");
            }
            printf("%d\n", next);
            if (next == 1) break;                  /* :1017 */
            token = next;                          /* :1036 */
        }
        l2z_runstate_free(s);
        l2z_weights_free(w);
    }
    return 0;
}
