// fm_constants.h — layout constants shared by the kernels (fm_kernels.h) and the library's pure host arithmetic
// (fmhip_host.h).  No HIP include: fmhip_host.cpp also compiles with plain g++ under AddressSanitizer / UBSan (CPU tests).
#pragma once
#include <stdint.h>

namespace fmhip {

constexpr int kRangeLen = 64;      // == FMHIP_RANGE_LEN
constexpr int kXcds = 8;           // L2 domains of an MI355X (workgroups are dispatched round-robin over them)
constexpr int kXSegs = 9;          // runs of one XCD's range list: up to 8 row bands + the share of the unplaced ranges
constexpr int kRowBands = 16;      // row bands of the band-affine placement: two per XCD, 2 MB of P each at 250k-row batches of Kp = 32
constexpr int kExtend = 16;        // a slot finishes a column that ends this close behind its range

}  // namespace fmhip
