"""CPU: the oracle's one-polynomial static-Huffman recoders (oracle/sc_oracle_bliss.c: orc_huff_row_encode / _decode) against the reference's own raw and
coded signatures of the same polynomial (tests/golden/huff_rows_ref.txt, made by `ref_driver huff_rows`: Falcon-512 / -1024 s2, ring-TESLA z | digest -
the same context signing the same message from the same CSPRNG seed with sc->coding_signature.type = NONE and = HUFFMAN_STATIC)."""
import numpy as np
import pytest

from tests.golden_io import load_golden
from tests import oracle_lib as O

CASES = [('falcon0', 4), ('falcon2', 2), ('tesla0', 4)]


@pytest.mark.parametrize('tag,count', CASES)
def test_recoders_match_the_reference(tag, count):
    H = load_golden('huff_rows_ref.txt')
    n, bits, tail, _ = (int(x) for x in H[f'{tag}_layout'])
    tab = O.huffman_table_for(H, bits)
    assert list(H[f'{tag}_verify']) == [0] * count                     # the reference verifies its own coded signatures
    for i in range(count):
        raw, coded = bytes(H[f'{tag}_raw{i}'].astype(np.uint8)), bytes(H[f'{tag}_coded{i}'].astype(np.uint8))
        assert len(raw) == (n * bits + 7) // 8 + tail and len(coded) < len(raw)
        rc, enc = O.huff_row_encode(raw, n, bits, tail, tab)
        assert rc == 0 and enc == coded, (tag, i)
        rc, dec = O.huff_row_decode(coded, n, bits, tail, tab)
        assert rc == 0 and dec == raw, (tag, i)
        rc, _ = O.huff_row_decode(coded[:len(coded) // 2], n, bits, tail, tab)
        assert rc == 1                                                   # a stream that ends early is reported
