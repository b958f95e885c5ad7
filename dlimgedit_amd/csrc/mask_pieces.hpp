// How the staging area of a mask request travels to the host and is copied out: pure host logic (no HIP), tested without a
// GPU through dlimg_amd_test_mask_pieces (tests/test_mask_pieces.py).
//   staging layout: mask i at offset sum of padded sizes of masks 0 .. i-1, padded to 256 bytes; optional IoU floats behind
//   pieces        : one mask in one piece (a piece costs a copy command and an event: 20 us for four of them); two masks or
//                   more in up to six pieces of about 1 MiB, so that the host's copy-out of piece i runs beside the
//                   transfer of piece i + 1 (five masks per call: 0.81 -> 0.67 ms)
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

namespace dlimg {

inline size_t padded_mask_bytes(size_t bytes) { return (bytes + 255) / 256 * 256; }

// end offsets of the pieces of a staging area of `total` bytes
inline std::vector<size_t> mask_piece_ends(size_t total) {
    constexpr size_t kPiece = 1024 * 1024;
    std::vector<size_t> ends;
    if (total == 0) return ends;
    const size_t pieces = total < 2 * kPiece ? 1 : std::min<size_t>(6, total / kPiece);
    const size_t piece = (total / pieces + 255) / 256 * 256;
    for (size_t a = 0; a < total; a += piece) ends.push_back(std::min(total, a + piece));
    return ends;
}

struct MaskCopy { int mask; size_t staging_offset; size_t mask_offset; size_t bytes; };

// What the host copies out once the piece [begin, end) has arrived: the parts of the masks (sizes in bytes, laid out as
// above) that lie in it.  `cursor` = first mask that is not finished yet and its staging offset, carried from piece to piece.
struct MaskCursor { int mask = 0; size_t offset = 0; };
inline std::vector<MaskCopy> mask_copies_in_piece(std::vector<size_t> const& sizes, size_t begin, size_t end, MaskCursor& cursor) {
    std::vector<MaskCopy> out;
    while (cursor.mask < (int)sizes.size()) {
        const size_t len = sizes[cursor.mask];
        const size_t a = std::max(begin, cursor.offset), b = std::min(end, cursor.offset + len);
        if (b > a) out.push_back(MaskCopy{cursor.mask, a, a - cursor.offset, b - a});
        if (cursor.offset + padded_mask_bytes(len) > end) break;           // the rest of this mask is in the next piece
        cursor.offset += padded_mask_bytes(len);
        ++cursor.mask;
    }
    return out;
}

}  // namespace dlimg
