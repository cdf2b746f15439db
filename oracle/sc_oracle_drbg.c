/*
 * sc_oracle_drbg.c - CPU restatement of the reference's default CSPRNG, the AES-256 CTR-DRBG behind create_csprng() and
 * sc->prng_ctx[0].  TEST INFRASTRUCTURE ONLY (checker for tests/, smoke() and bench.py's cpu_baseline leg).
 *
 * Follows
 *   src/utils/crypto/ctr_drbg.c:37-197      create (zeroed context: key = 0, counter = 0), reseed/update, ctr_drbg_update
 *   src/utils/crypto/prng_get_func.c:108-119, 159-193   the circular user-entropy reader, get_random_64_aes (1024-byte buffer)
 *   src/utils/crypto/prng.c:95-143, 963-993, 1050-1107  the 4096-word pool (two words per 64-bit draw, HIGH word first),
 *                                                        prng_32 / prng_64, prng_mem (whole 64-byte blocks per call)
 *   src/utils/arith/module_lwe.c:914-940    create_csprng: entropy = the seed, read circularly; the nonce is not used by this DRBG
 * The block cipher is the reference's third-party AES (src/utils/crypto/aes, Gladman): plain FIPS-197 AES-256 encryption,
 * restated here from the standard (S-box generated from the field inverse + affine map at first use).
 *
 * Instantiation: counter 1,2,3 (each replicated into the four little-endian words of a block) are encrypted under the zero
 * key into bytes[32..48), [16..32), [0..16); then 4 seed bytes -> ctr, the next 32 (circular) -> key; key ^= bytes[12..44);
 * counter = 3 ^ ctr; new key schedule.  Generation: block i of a 1024-byte buffer = AES_key(counter++ replicated x4).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "sc_oracle.h"

static uint8_t sbox[256];
static int sbox_ready;

static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static uint8_t gmul(uint8_t a, uint8_t b) { uint8_t r = 0; while (b) { if (b & 1) r ^= a; a = xtime(a); b >>= 1; } return r; }

static void sbox_init(void)
{
    if (sbox_ready) return;
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gmul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; i++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        sbox[x] = s ^ 0x63;
    }
    sbox_ready = 1;
}

typedef struct { uint8_t rk[15][16]; } aes256_t;

static void aes256_key(aes256_t *a, const uint8_t key[32])
{
    sbox_init();
    uint8_t w[60][4];
    memcpy(w, key, 32);
    uint8_t rc = 1;
    for (int i = 8; i < 60; i++) {
        uint8_t t[4]; memcpy(t, w[i - 1], 4);
        if (i % 8 == 0) {
            const uint8_t t0 = t[0];
            t[0] = sbox[t[1]] ^ rc; t[1] = sbox[t[2]]; t[2] = sbox[t[3]]; t[3] = sbox[t0];
            rc = xtime(rc);
        } else if (i % 8 == 4) {
            for (int j = 0; j < 4; j++) t[j] = sbox[t[j]];
        }
        for (int j = 0; j < 4; j++) w[i][j] = w[i - 8][j] ^ t[j];
    }
    memcpy(a->rk, w, 240);
}

static void aes256_enc(const aes256_t *a, const uint8_t in[16], uint8_t out[16])
{
    uint8_t s[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ a->rk[0][i];
    for (int r = 1; r <= 14; r++) {
        uint8_t t[16];
        for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) t[4 * c + row] = sbox[s[4 * ((c + row) & 3) + row]];   /* SubBytes + ShiftRows */
        if (r < 14)
            for (int c = 0; c < 4; c++) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c]     = (uint8_t)(xtime(a0) ^ xtime(a1) ^ a1 ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ xtime(a1) ^ xtime(a2) ^ a2 ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xtime(a2) ^ xtime(a3) ^ a3);
                s[4 * c + 3] = (uint8_t)(xtime(a0) ^ a0 ^ a1 ^ a2 ^ xtime(a3));
            }
        else memcpy(s, t, 16);
        for (int i = 0; i < 16; i++) s[i] ^= a->rk[r][i];
    }
    memcpy(out, s, 16);
}

void orc_aes256_ecb(const uint8_t key[32], const uint8_t in[16], uint8_t out[16])
{
    aes256_t a; aes256_key(&a, key); aes256_enc(&a, in, out);
}

struct orc_drbg {
    aes256_t aes;
    uint32_t counter;
    uint8_t buf[1024]; int cnt64;          /* get_random_64_aes: index of the next 64-bit word, 128 = empty */
    uint32_t pool[4096]; int rd, bits;     /* prng.c pool */
};

/* AES(counter value replicated into the four little-endian words of a block) */
static void enc_ctr(const orc_drbg_t *d, uint32_t c, uint8_t out[16])
{
    uint8_t blk[16];
    for (int w = 0; w < 4; w++) { blk[4 * w] = (uint8_t)c; blk[4 * w + 1] = (uint8_t)(c >> 8); blk[4 * w + 2] = (uint8_t)(c >> 16); blk[4 * w + 3] = (uint8_t)(c >> 24); }
    aes256_enc(&d->aes, blk, out);
}
static void ctr_block(orc_drbg_t *d, uint8_t out[16]) { enc_ctr(d, d->counter++, out); }       /* ctr_drbg.c:182-188 */

orc_drbg_t *orc_drbg_create(const uint8_t *seed, size_t len)
{
    if (!seed || len == 0) return NULL;
    orc_drbg_t *d = calloc(1, sizeof *d);
    uint8_t key[32] = { 0 }, bytes[48];
    aes256_key(&d->aes, key);
    d->counter = 0;
    for (int block = 3; block;) { d->counter++; block--; enc_ctr(d, d->counter, bytes + 16 * block); }     /* ctr_drbg.c:104-128 */
    size_t idx = 0;
    uint8_t ctr[4];
    for (int i = 0; i < 4; i++) { ctr[i] = seed[idx++]; if (idx == len) idx = 0; }
    for (int i = 0; i < 32; i++) { key[i] = seed[idx++]; if (idx == len) idx = 0; }
    for (int i = 0; i < 32; i++) key[i] ^= bytes[12 + i];
    d->counter ^= (uint32_t)ctr[0] | ((uint32_t)ctr[1] << 8) | ((uint32_t)ctr[2] << 16) | ((uint32_t)ctr[3] << 24);
    aes256_key(&d->aes, key);
    d->cnt64 = 128;
    d->bits = 0; d->rd = 0;
    return d;
}

void orc_drbg_destroy(orc_drbg_t *d) { free(d); }

static uint64_t next64(orc_drbg_t *d)
{
    if (d->cnt64 == 128) { d->cnt64 = 0; for (int i = 0; i < 1024; i += 16) ctr_block(d, d->buf + i); }
    uint64_t v; memcpy(&v, d->buf + 8 * d->cnt64++, 8);       /* little-endian host, as the reference's union read */
    return v;
}

/* prng.c:1050-1107 */
void orc_drbg_mem(orc_drbg_t *d, uint8_t *mem, int32_t length)
{
    int32_t blocks = (length + 63) >> 6;
    while (blocks--) {
        uint8_t data[64];
        for (int k = 0; k < 8; k++) { const uint64_t v = next64(d); memcpy(data + 8 * k, &v, 8); }
        memcpy(mem, data, length >= 64 ? 64 : (size_t)length);
        length -= 64; mem += 64;
    }
}

/* prng.c:95-143, 981-993 */
uint32_t orc_drbg_32(orc_drbg_t *d)
{
    if (d->bits == 0) {
        for (int i = 0; i < 2048; i++) { const uint64_t v = next64(d); d->pool[2 * i] = (uint32_t)(v >> 32); d->pool[2 * i + 1] = (uint32_t)v; }
        d->rd = 0; d->bits = 4096 * 32;
    }
    const uint32_t w = d->pool[d->rd];
    d->bits -= 32; d->rd = (d->rd + 1) & 4095;
    return w;
}

/* prng.c:963-976 */
uint64_t orc_drbg_64(orc_drbg_t *d) { const uint64_t hi = orc_drbg_32(d); return (hi << 32) | orc_drbg_32(d); }

/* uniform_random_ring_q_csprng (src/utils/arith/module_lwe.c:519-547, n = 256) on create_csprng(rho): polynomial p is the p-th
 * prng_mem(512): 256 little-endian 16-bit words masked to q_bits, minus q when >= q */
void orc_expand_matrix(const uint8_t *rho, size_t rho_len, int npoly, int32_t q, int q_bits, int32_t *out)
{
    orc_drbg_t *d = orc_drbg_create(rho, rho_len);
    const int32_t mask = (1 << q_bits) - 1;
    for (int p = 0; p < npoly; p++) {
        uint8_t b[512];
        orc_drbg_mem(d, b, 512);
        for (int i = 0; i < 256; i++) { int32_t v = (int32_t)((uint32_t)b[2 * i] | ((uint32_t)b[2 * i + 1] << 8)) & mask; v -= (v >= q) * q; out[p * 256 + i] = v; }
    }
    orc_drbg_destroy(d);
}
