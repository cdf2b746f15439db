/* sc_oracle_keccak.h - internal to oracle/: the reference's sponge.  TEST INFRASTRUCTURE ONLY. */
#ifndef SC_ORACLE_KECCAK_H
#define SC_ORACLE_KECCAK_H
#include <stddef.h>
#include <stdint.h>
typedef struct { uint64_t s[25]; unsigned rate, pos; } shake_t;
extern int orc_keccak_standard;
void orc_keccakf(uint64_t st[25]);
void shake_init(shake_t *c, int bits);
void shake_absorb(shake_t *c, const uint8_t *in, size_t len);
void shake_final(shake_t *c);
void shake_squeeze(shake_t *c, uint8_t *out, size_t len);
void sha3_init(shake_t *c, int mdlen);
void sha3_final(shake_t *c, uint8_t *md, int mdlen);
#endif
