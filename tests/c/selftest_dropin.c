/*
 * selftest_dropin.c - a C program written against eddsa.h only (the reference's public header
 * surface, reference lib/eddsa.h:44-113), linked against libeddsa_amd.so instead of libeddsa.so.
 * It performs the checks of the reference's own selftests on the golden tables:
 *   test/selftest-x25519.c:15-48        x25519 and DH on every table entry
 *   test/selftest-ed25519.c:30-80       genpub, sign (deterministic), verify, and the obsolete names
 *   test/selftest-x25519_base.c:19-45   x25519_base(x) == x25519(x, 9)
 *   test/selftest-convert.c:20-80       x25519_base(sk->x) == pk->x(genpub(sk))
 * plus the batched entry points of eddsa_amd.h on the same data.
 * usage: selftest_dropin <x25519_table.bin> <ed25519_table.bin> <ed25519_msgs.bin>
 * exit status 0 = all checks passed.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eddsa.h"
#include "eddsa_amd.h"

static uint8_t *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *p = malloc(*len ? *len : 1);
    if (fread(p, 1, *len, f) != *len) { perror("fread"); exit(2); }
    fclose(f);
    return p;
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "selftest_dropin: " __VA_ARGS__); fputc('\n', stderr); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s x25519_table.bin ed25519_table.bin ed25519_msgs.bin\n", argv[0]); return 2; }
    size_t xl, el, ml;
    uint8_t *xt = slurp(argv[1], &xl), *et = slurp(argv[2], &el), *msgs = slurp(argv[3], &ml);
    const size_t nx = xl / 96, ne = el / 128;
    CHECK(nx == 1024 && ne == 1024, "unexpected table sizes");
    uint8_t out[64];

    /* selftest-x25519.c: fields are point, scalar, result; every 16th entry through the single-item API */
    for (size_t i = 0; i < nx; i += 16) {
        const uint8_t *pt = xt + 96 * i, *sc = pt + 32, *res = pt + 64;
        x25519(out, sc, pt);
        CHECK(memcmp(out, res, 32) == 0, "x25519 entry %zu", i + 1);
        DH(out, sc, pt);
        CHECK(memcmp(out, res, 32) == 0, "DH entry %zu", i + 1);
    }
    {   /* the whole table through the batched entry point */
        uint8_t *sc = malloc(32 * nx), *pt = malloc(32 * nx), *res = malloc(32 * nx);
        for (size_t i = 0; i < nx; i++) { memcpy(pt + 32 * i, xt + 96 * i, 32); memcpy(sc + 32 * i, xt + 96 * i + 32, 32); }
        CHECK(x25519_batch(res, sc, pt, nx) == 0, "x25519_batch failed");
        for (size_t i = 0; i < nx; i++) CHECK(memcmp(res + 32 * i, xt + 96 * i + 64, 32) == 0, "x25519_batch entry %zu", i + 1);
        /* selftest-x25519_base.c */
        uint8_t nine[32] = { 9 };
        for (size_t i = 0; i < nx; i++) memcpy(pt + 32 * i, nine, 32);
        uint8_t *b1 = malloc(32 * nx);
        CHECK(x25519_base_batch(b1, sc, nx) == 0 && x25519_batch(res, sc, pt, nx) == 0, "x25519_base_batch failed");
        CHECK(memcmp(b1, res, 32 * nx) == 0, "x25519_base(x) != x25519(x, 9)");
        x25519_base(out, sc);
        CHECK(memcmp(out, b1, 32) == 0, "x25519_base single");
        /* selftest-convert.c */
        uint8_t *xs = malloc(32 * nx), *pk = malloc(32 * nx), *xp = malloc(32 * nx);
        CHECK(sk_ed25519_to_x25519_batch(xs, sc, nx) == 0 && x25519_base_batch(b1, xs, nx) == 0, "sk conversion failed");
        CHECK(ed25519_genpub_batch(pk, sc, nx) == 0 && pk_ed25519_to_x25519_batch(xp, pk, nx) == 0, "pk conversion failed");
        CHECK(memcmp(b1, xp, 32 * nx) == 0, "x25519_base(sk->x) != pk->x(genpub(sk))");
        sk_ed25519_to_x25519(out, sc); eddsa_sk_eddsa_to_dh(out + 32, sc);
        CHECK(memcmp(out, xs, 32) == 0 && memcmp(out + 32, xs, 32) == 0, "sk_ed25519_to_x25519 single");
        pk_ed25519_to_x25519(out, pk); eddsa_pk_eddsa_to_dh(out + 32, pk);
        CHECK(memcmp(out, xp, 32) == 0 && memcmp(out + 32, xp, 32) == 0, "pk_ed25519_to_x25519 single");
        free(sc); free(pt); free(res); free(b1); free(xs); free(pk); free(xp);
    }

    /* selftest-ed25519.c: entry i = sk | pk | sig, message of i bytes at offset i(i-1)/2 of msgs */
    CHECK(ml == ne * (ne - 1) / 2, "message blob has the wrong size");
    for (size_t i = 0; i < ne; i += 31) {
        const uint8_t *sk = et + 128 * i, *pk = sk + 32, *sig = sk + 64, *m = msgs + i * (i - 1) / 2;
        ed25519_genpub(out, sk);
        CHECK(memcmp(out, pk, 32) == 0, "generating ed25519 public key number %zu", i + 1);
        ed25519_sign(out, sk, pk, m, i);
        CHECK(memcmp(out, sig, 64) == 0, "generating ed25519 signature number %zu", i + 1);
        CHECK(ed25519_verify(sig, pk, m, i), "verifying ed25519 signature number %zu", i + 1);
        eddsa_genpub(out, sk);
        CHECK(memcmp(out, pk, 32) == 0, "eddsa_genpub number %zu", i + 1);
        eddsa_sign(out, sk, pk, m, i);
        CHECK(memcmp(out, sig, 64) == 0, "eddsa_sign number %zu", i + 1);
        CHECK(eddsa_verify(sig, pk, m, i), "eddsa_verify number %zu", i + 1);
        if (i) { memcpy(out, sig, 64); out[5] ^= 4; CHECK(!ed25519_verify(out, pk, m, i), "forged signature %zu accepted", i + 1); }
    }
    {   /* the whole table through the batched entry points, ragged messages */
        uint64_t *off = malloc(sizeof(uint64_t) * (ne + 1));
        uint8_t *sk = malloc(32 * ne), *pk = malloc(32 * ne), *sig = malloc(64 * ne), *chk = malloc(64 * ne), *ok = malloc(ne);
        for (size_t i = 0; i <= ne; i++) off[i] = (uint64_t)(i * (i - 1) / 2 + (i ? 0 : 0));
        off[0] = 0;
        for (size_t i = 0; i < ne; i++) { memcpy(sk + 32 * i, et + 128 * i, 32); memcpy(pk + 32 * i, et + 128 * i + 32, 32); memcpy(sig + 64 * i, et + 128 * i + 64, 64); }
        CHECK(ed25519_genpub_batch(chk, sk, ne) == 0 && memcmp(chk, pk, 32 * ne) == 0, "ed25519_genpub_batch");
        CHECK(ed25519_sign_batch(chk, sk, pk, msgs, off, 0, ne) == 0 && memcmp(chk, sig, 64 * ne) == 0, "ed25519_sign_batch");
        CHECK(ed25519_verify_batch(ok, sig, pk, msgs, off, 0, ne) == 0, "ed25519_verify_batch failed");
        for (size_t i = 0; i < ne; i++) CHECK(ok[i] == 1, "ed25519_verify_batch rejected number %zu", i + 1);
        {   /* record form: the public keys signed as 32-byte messages, one 136-byte record per item
             * (4 bytes of header, sig, pub, msg, 4 bytes of trailer), every third signature forged */
            const size_t stride = 136;
            uint8_t *rec = calloc(ne, stride);
            CHECK(ed25519_sign_batch(chk, sk, pk, pk, NULL, 32, ne) == 0, "ed25519_sign_batch (fixed length)");
            for (size_t i = 0; i < ne; i++) {
                uint8_t *r = rec + stride * i;
                memcpy(r + 4, chk + 64 * i, 64); memcpy(r + 68, pk + 32 * i, 32); memcpy(r + 100, pk + 32 * i, 32);
                if (i % 3 == 2) r[4 + 40] ^= 1;
            }
            CHECK(ed25519_verify_records(ok, rec, stride, 4, 68, 100, 32, ne) == 0, "ed25519_verify_records failed");
            for (size_t i = 0; i < ne; i++) CHECK(ok[i] == (i % 3 != 2), "ed25519_verify_records verdict %zu", i + 1);
            CHECK(ed25519_verify_records(ok, rec, stride, 80, 68, 100, 32, ne) != 0, "a signature outside the record was accepted as a layout");
            free(rec);
        }
        free(off); free(sk); free(pk); free(sig); free(chk); free(ok);
    }
    eddsa_amd_shutdown();
    free(xt); free(et); free(msgs);
    printf("selftest_dropin: ok (%zu x25519 vectors, %zu ed25519 vectors)\n", nx, ne);
    return 0;
}
