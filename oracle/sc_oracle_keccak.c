/*
 * sc_oracle_keccak.c - the reference's sponge (SHAKE128/256 and SHA3-512 framing around ITS permutation).
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h).
 */
#include "sc_oracle_keccak.h"
#include <string.h>

/* The reference's xof layer (src/utils/crypto/xof.c over sha3/tinysha3.c) is SHAKE128/256 sponge
 * framing (rate 168/136, pad 0x1F .. 0x80, incremental absorb/squeeze) around a permutation that is
 * NOT Keccak-f[1600]: the library is always compiled with -DSHA3_UNROLLED
 * (src/utils/crypto/Makefile.am:94) and the unrolled chi step (sha3/tinysha3.c:186-235) walks the
 * five "rows" at offsets 0,1,2,3,4 instead of 0,5,10,15,20 - five overlapping, sequentially applied
 * updates of st[0..8]; st[9..24] never see chi.  theta, rho-pi and iota are the standard ones.  Every
 * hash the schemes take goes through it, so signatures are only reproducible with the same quirk.
 * orc_keccak_standard = 1 switches to the FIPS 202 chi (used only to validate this file's sponge
 * against hashlib). */
int orc_keccak_standard = 0;

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
    0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
    0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
    0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL };
static const int ROTC[24] = { 1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44 };
static const int PILN[24] = { 10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1 };
#define ROL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void orc_keccakf(uint64_t st[25])
{
    uint64_t bc[5], t;
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) { t = bc[(i + 4) % 5] ^ ROL64(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
        t = st[1];
        for (int i = 0; i < 24; i++) { int j = PILN[i]; bc[0] = st[j]; st[j] = ROL64(t, ROTC[i]); t = bc[0]; }
        for (int jj = 0; jj < 5; jj++) {
            const int j = orc_keccak_standard ? 5 * jj : jj;          /* tinysha3.c:186-235: j = 0,1,2,3,4 */
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[r];
    }
}
void shake_init(shake_t *c, int bits) { memset(c, 0, sizeof *c); c->rate = 200 - 2 * (unsigned)bits / 8; }
void shake_absorb(shake_t *c, const uint8_t *in, size_t len)
{
    for (size_t i = 0; i < len; i++) {
        c->s[c->pos >> 3] ^= (uint64_t)in[i] << (8 * (c->pos & 7));
        if (++c->pos == c->rate) { orc_keccakf(c->s); c->pos = 0; }
    }
}
void shake_final(shake_t *c)
{
    c->s[c->pos >> 3] ^= (uint64_t)0x1F << (8 * (c->pos & 7));
    c->s[(c->rate - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((c->rate - 1) & 7));
    orc_keccakf(c->s);
    c->pos = 0;
}
void shake_squeeze(shake_t *c, uint8_t *out, size_t len)
{
    for (size_t i = 0; i < len; i++) {
        if (c->pos == c->rate) { orc_keccakf(c->s); c->pos = 0; }
        out[i] = (uint8_t)(c->s[c->pos >> 3] >> (8 * (c->pos & 7)));
        c->pos++;
    }
}

void orc_shake(int bits, const uint8_t *in, size_t inlen, uint8_t *out, size_t outlen)
{
    shake_t c; shake_init(&c, bits); shake_absorb(&c, in, inlen); shake_final(&c); shake_squeeze(&c, out, outlen);
}


/* tinysha3_final (sha3/tinysha3.c:334-348): fixed-output SHA-3, pad 0x06 .. 0x80, rate 200 - 2*mdlen */
void sha3_init(shake_t *c, int mdlen) { memset(c, 0, sizeof *c); c->rate = 200 - 2 * (unsigned)mdlen; }
void sha3_final(shake_t *c, uint8_t *md, int mdlen)
{
    c->s[c->pos >> 3] ^= (uint64_t)0x06 << (8 * (c->pos & 7));
    c->s[(c->rate - 1) >> 3] ^= (uint64_t)0x80 << (8 * ((c->rate - 1) & 7));
    orc_keccakf(c->s);
    for (int i = 0; i < mdlen; i++) md[i] = (uint8_t)(c->s[i >> 3] >> (8 * (i & 7)));
}

void orc_sha3(int mdlen, const uint8_t *in, size_t inlen, uint8_t *md)
{
    shake_t c; sha3_init(&c, mdlen); shake_absorb(&c, in, inlen); sha3_final(&c, md, mdlen);
}
