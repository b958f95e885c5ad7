"""Host logic of the mask transfer (csrc/mask_pieces.hpp) through the GPU-free hook dlimg_amd_test_mask_pieces: the pieces a
staging area travels in and what the host copies out of each -- every byte of every mask exactly once, from where the
post-processing kernel put it."""
import numpy as np
import pytest

from dlimgedit_amd import api

MIB = 1024 * 1024


def pad(n):
    return (n + 255) // 256 * 256


def reassemble(sizes, extra=0):
    """Plays the transfer on a staging area whose byte at offset o is o % 251: returns the masks as the host would see them."""
    ends, copies = api.ext.mask_pieces(sizes, extra)
    total = sum(pad(s) for s in sizes) + extra
    staging = (np.arange(total, dtype=np.int64) % 251).astype(np.uint8)
    masks = [np.full(s, 255, np.uint8) for s in sizes]
    written = [np.zeros(s, np.int32) for s in sizes]
    arrived = 0
    for piece, mask, so, mo, n in copies:
        assert so + n <= ends[piece] and so >= (ends[piece - 1] if piece else 0), "a copy reads outside its piece"
        arrived = max(arrived, ends[piece])
        masks[mask][mo:mo + n] = staging[so:so + n]
        written[mask][mo:mo + n] += 1
    return ends, masks, written, staging


@pytest.mark.parametrize("sizes", [
    [MIB], [MIB, MIB], [MIB] * 5, [MIB] * 8, [MIB] * 16, [1024 * 683], [1024 * 683] * 3, [7], [7, 1000003, 2], [MIB, 3 * MIB + 5, 100],
    [640 * 480] * 7, [1800 * 1200] * 5,
])
@pytest.mark.parametrize("extra", [0, 16, 64])
def test_every_byte_of_every_mask_is_copied_exactly_once(sizes, extra):
    ends, masks, written, staging = reassemble(sizes, extra)
    off = 0
    for s, m, w in zip(sizes, masks, written):
        assert (w == 1).all()
        assert np.array_equal(m, staging[off:off + s])
        off += pad(s)
    assert ends[-1] == off + extra and ends == sorted(ends)


def test_one_mask_travels_in_one_piece_and_many_in_at_most_six():
    assert len(api.ext.mask_pieces([MIB])[0]) == 1
    assert len(api.ext.mask_pieces([MIB, MIB])[0]) == 2
    assert len(api.ext.mask_pieces([MIB] * 5)[0]) == 5
    ends, _ = api.ext.mask_pieces([MIB] * 16)
    assert len(ends) == 6 and all(e % 256 == 0 for e in ends[:-1])


def test_no_masks_no_pieces():
    assert api.ext.mask_pieces([]) == ([], [])
