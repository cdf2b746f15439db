"""CPU: the oracle's BLISS-B sign/verify (oracle/sc_oracle_bliss.c) against signatures the reference itself
produced from the same (key, message, random tape) (tests/golden/bliss_ref.txt, generator oracle/ref_driver.c)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O


@pytest.fixture(scope='module')
def G():
    return load_golden('bliss_ref.txt')


def cases(G, tag='bliss1'):
    i = 0
    while f'{tag}_msg{i}' in G:
        yield (i, bytes(G[f'{tag}_msg{i}'].astype(np.uint8)), G[f'{tag}_tape{i}'], bytes(G[f'{tag}_sig{i}'].astype(np.uint8)),
               int(G[f'{tag}_trials'][i]))
        i += 1


def test_sign_reproduces_reference_signatures(G):
    key = O.BlissKey.from_golden(G, 'bliss1')
    seen = set()
    for i, m, tape, sig, trials in cases(G):
        rc, got, tr, used = O.bliss_sign(key, m, tape)
        assert rc == 0 and got == sig, i
        assert tr == trials and used == len(tape)          # same attempts, same number of CSPRNG words consumed
        seen.add(trials)
    assert max(seen) >= 3                                   # a case that resampled after two rejections is covered


def test_tape_underrun_is_reported(G):
    key = O.BlissKey.from_golden(G, 'bliss1')
    _, m, tape, _, _ = next(cases(G))
    rc, _, _, _ = O.bliss_sign(key, m, tape[:1000])
    assert rc == 2


def test_verify_matches_reference_verdicts(G):
    key = O.BlissKey.from_golden(G, 'bliss1')
    for i, m, _, sig, _ in cases(G):
        assert O.bliss_verify(key, m, sig) == int(G['bliss1_verify'][2 * i]) == 0
        bad = bytearray(sig); bad[len(bad) // 3] ^= 0x04
        assert O.bliss_verify(key, m, bytes(bad)) == int(G['bliss1_verify'][2 * i + 1]) == 1
        assert O.bliss_verify(key, m + b'!', sig) == 1


def test_keygen_matches_reference(G):
    """bliss_b_keygen on the reference's own CSPRNG words: f, g, a equal, every word consumed; case 3 had to draw f again"""
    par, nz = G['bliss1_params'], G['bliss1_kg_nz']
    atts = []
    for k in range(4):
        tape = G[f'bliss1_kg_tape{k}']
        rc, f, g, a, used, att = O.bliss_keygen(par, nz, tape)
        assert rc == 0 and used == len(tape), k
        assert np.array_equal(f, G[f'bliss1_kg_f{k}']) and np.array_equal(g, G[f'bliss1_kg_g{k}']) and np.array_equal(a, G[f'bliss1_kg_a{k}']), k
        atts.append(att)
    assert atts[:3] == [1, 1, 1] and atts[3] > 1
    assert O.bliss_keygen(par, nz, G['bliss1_kg_tape3'][:500])[0] == 2


@pytest.mark.parametrize('tag', ['bliss3', 'bliss4'])
def test_other_parameter_sets_match_reference(G, tag):
    """BLISS-B-III and -IV (other kappa, sigma, d, p and norm bounds): the reference's signatures, attempts, word consumption and verdicts"""
    key = O.BlissKey.from_golden(G, tag)
    for i, m, tape, sig, trials in cases(G, tag):
        rc, got, tr, used = O.bliss_sign(key, m, tape)
        assert rc == 0 and got == sig and tr == trials and used == len(tape), i
        assert O.bliss_verify(key, m, sig) == int(G[f'{tag}_verify'][2 * i]) == 0
        bad = bytearray(sig); bad[len(bad) // 3] ^= 0x04
        assert O.bliss_verify(key, m, bytes(bad)) == int(G[f'{tag}_verify'][2 * i + 1])


@pytest.mark.parametrize('tag', ['blissh1', 'blissh4', 'blissh2'])
def test_huffman_coded_signatures_match_the_reference(tag):
    """SC_FLAG_0_ENTROPY_HUFFMAN: the reference's coded signatures (tests/golden/bliss_huff_ref.txt) = the raw signature of the same (key, message,
    tape) passed through the static-table coder; decoding gives the raw signature back, which verifies; the reference's verdicts on the coded
    signatures with a flipped bit are reproduced after decoding"""
    H = load_golden('bliss_huff_ref.txt')
    key = O.BlissKey.from_golden(H, tag)
    tabs = O.huffman_tables(H, int(key.par[9]), int(key.par[10]))
    raw_len = (key.n * (int(key.par[9]) + int(key.par[10])) + int(key.par[6]) * int(key.par[3]) + 7) // 8
    for i, m, tape, sig, trials in cases(H, tag):
        rc, raw, tr, used = O.bliss_sign(key, m, tape)
        assert rc == 0 and tr == trials and used == len(tape), (tag, i)
        rc, coded = O.bliss_sig_to_huffman(key, raw, tabs)
        assert rc == 0 and coded == sig, (tag, i)
        assert len(coded) < raw_len
        rc, back = O.bliss_sig_from_huffman(key, sig, tabs)
        assert rc == 0 and back == raw, (tag, i)
        assert O.bliss_verify(key, m, back) == int(H[f'{tag}_verify'][2 * i]) == 0
        bad = bytearray(sig); bad[len(bad) // 3] ^= 0x04
        rc, braw = O.bliss_sig_from_huffman(key, bytes(bad), tabs)
        assert (1 if rc else O.bliss_verify(key, m, braw)) == int(H[f'{tag}_verify'][2 * i + 1]) == 1
        rc, _ = O.bliss_sig_from_huffman(key, sig[:len(sig) // 2], tabs)
        assert rc == 1                                          # a stream that ends early


@pytest.mark.parametrize('tag', ['blissb1', 'blissb4'])
def test_arithmetic_coded_signatures_match_the_reference(tag):
    """SC_FLAG_0_ENTROPY_BAC: the reference's coded signatures (tests/golden/bliss_bac_ref.txt) = the raw signature of the same (key, message, tape)
    with t and z passed through the arithmetic coder of src/utils/entropy/bac.c under the two distributions the reference built (recorded as data);
    decoding gives the raw signature back; the reference's verdicts on coded signatures with a flipped bit are reproduced after decoding"""
    H = load_golden('bliss_bac_ref.txt')
    key = O.BlissKey.from_golden(H, tag)
    dists = O.bac_dists(H, tag)
    assert dists[0].size == 1 << int(key.par[9]) and dists[1].size == 1 << int(key.par[10])
    raw_len = (key.n * (int(key.par[9]) + int(key.par[10])) + int(key.par[6]) * int(key.par[3]) + 7) // 8
    for i, m, tape, sig, trials in cases(H, tag):
        rc, raw, tr, used = O.bliss_sign(key, m, tape)
        assert rc == 0 and tr == trials and used == len(tape), (tag, i)
        rc, coded = O.bliss_sig_to_bac(key, raw, dists)
        assert rc == 0 and coded == sig, (tag, i)
        assert len(coded) < raw_len
        rc, back = O.bliss_sig_from_bac(key, sig, dists)
        assert rc == 0 and back == raw, (tag, i)
        assert O.bliss_verify(key, m, back) == int(H[f'{tag}_verify'][2 * i]) == 0
        bad = bytearray(sig); bad[len(bad) // 3] ^= 0x04
        rc, braw = O.bliss_sig_from_bac(key, bytes(bad), dists)
        assert (1 if rc else O.bliss_verify(key, m, braw)) == int(H[f'{tag}_verify'][2 * i + 1]) == 1
        rc, _ = O.bliss_sig_from_bac(key, sig[:len(sig) // 2], dists)
        assert rc == 1                                          # a stream that ends early


def test_arithmetic_coder_distributions():
    """gauss_freq_bac_64 restated: long double exp() arithmetic, so - like the CDT tables - entries may differ in their last units between hosts;
    against the fixture made on the dev container's host: equal within 2^-40 relative, and bit for bit on that host"""
    H = load_golden('bliss_bac_ref.txt')
    for tag in ('blissb1', 'blissb4'):
        key = O.BlissKey.from_golden(H, tag)
        sigma = float(np.asarray(H[f'{tag}_params'], dtype=np.int32)[12:13].view(np.float32)[0])
        want = O.bac_dists(H, tag)
        for w, (sg, bits) in zip(want, ((sigma, int(key.par[9])), (0.7, int(key.par[10])))):
            got = O.bac_dist(sg, bits)
            assert got[0] == 0 and w[0] == 0
            rel = np.abs(got[1:].astype(np.float64) - w[1:].astype(np.float64)) / np.maximum(w[1:].astype(np.float64), 1.0)
            assert rel.max() < 2.0 ** -40, (tag, bits, rel.max())


def test_huffman_tables_are_prefix_codes():
    H = load_golden('bliss_huff_ref.txt')
    for t in range(2, 7):
        code, bits = H[f'huff{t}_code'].astype(np.int64) & 0xFFFFFFFF, H[f'huff{t}_bits']
        assert len(code) == int(H[f'huff{t}_meta'][0]) == 1 << t and int(bits.max()) == int(H[f'huff{t}_meta'][1])
        words = [format(int(c), 'b').zfill(int(b)) for c, b in zip(code, bits)]
        assert all(len(w) == int(b) for w, b in zip(words, bits))
        assert not any(a != b and b.startswith(a) for a in words for b in words)
        assert abs(sum(2.0 ** -len(w) for w in words) - 1.0) < 1e-12        # complete: every bit string decodes


def test_tape_bookkeeping_of_the_library_matches_the_reference():
    """host logic of the engine (no device): sc_hip_bliss_tape_words(n, attempts) - the reservoir bookkeeping that lets the signer place a message's later
    pairs of attempts on its tape - equals the number of CSPRNG words the reference drew for a signature of that many attempts (the fixtures' tapes are cut
    to exactly what safecrypto_sign consumed) and what the oracle consumes on long seeded tapes"""
    import ctypes
    from libsafecrypto_amd import engine
    lib = ctypes.CDLL(engine.LIB_PATH)
    lib.sc_hip_bliss_tape_words.argtypes = [ctypes.c_int, ctypes.c_uint]; lib.sc_hip_bliss_tape_words.restype = ctypes.c_size_t
    seen = set()
    for fx, tags in (('bliss_ref.txt', ('bliss1', 'bliss3', 'bliss4')), ('bliss_huff_ref.txt', ('blissh1', 'blissh4', 'blissh2'))):
        F = load_golden(fx)
        for tag in tags:
            n = int(F[f'{tag}_params'][0])
            for i, m, tape, sig, trials in cases(F, tag):
                assert lib.sc_hip_bliss_tape_words(n, trials) == len(tape), (tag, i, trials)
                seen.add(trials)
    assert max(seen) >= 4 and min(seen) == 1
    G = load_golden('bliss_ref.txt')
    key = O.BlissKey.from_golden(G, 'bliss4')
    rng = np.random.default_rng(3)
    deep = 0
    for _ in range(60):
        tape = rng.integers(0, 2**32, size=12 * 2048 + 100, dtype=np.uint64).astype(np.uint32)
        rc, _, tr, used = O.bliss_sign(key, bytes(rng.integers(0, 256, size=20, dtype=np.uint8)), tape)
        assert rc == 0 and lib.sc_hip_bliss_tape_words(key.n, tr) == used
        deep += tr >= 3
    assert deep >= 5
    assert lib.sc_hip_bliss_tape_words(512, 0) == 0 and lib.sc_hip_bliss_tape_words(0, 3) == 0
