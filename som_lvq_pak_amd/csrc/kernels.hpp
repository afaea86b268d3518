// kernels.hpp -- gfx950 (MI355X, CDNA4) device code of the SOM/LVQ engine.
//
// Numerics contract (bit-exact with the reference's CPU code): distances are the
// left-to-right fp32 sum of (c_i - x_i)^2 with a separate rounding after the
// subtract, the multiply and the add (reference lvq_pak.c:70-71); updates are
// c = c + alpha*(x - c) with three roundings (lvq_pak.c:348-349).  This file is
// compiled with -ffp-contract=off and must contain no fma/mac in those chains
// (checked on the ISA by tests/test_build.py).
//
// Data layout in HBM ("row-group tiles"): the codebook is stored as
//     tile[g][q][lane][4]    g = row / 64, lane = row % 64, q = dim / 4
// i.e. for every group of 64 consecutive code rows and every chunk of 4 dims, the
// 64 float4 of that chunk are contiguous (1 KiB).  One lane of a wavefront owns one
// code row, walks its dims in order (which the sequential-sum contract needs) and
// every wave-level load/store is one fully coalesced 1 KiB access.  Rows >= n and
// dims >= d are zero padding (adding +0.0f to a sum of squares is exact).
#pragma once

// The kernels live in kernels/*.hpp, one file per stage of the path; each includes its predecessor,
// so this umbrella only needs the last one.  K-numbers in comments and in DESIGN.md refer to the
// section banners inside those files.
#include "kernels/common.hpp"
#include "kernels/gauss_rate.hpp"
#include "kernels/layout.hpp"
#include "kernels/scan_exact.hpp"
#include "kernels/prefilter_mfma.hpp"
#include "kernels/scan_masked.hpp"
#include "kernels/som_update.hpp"
#include "kernels/som_update_gemm.hpp"
#include "kernels/som_online.hpp"
#include "kernels/rerank.hpp"
#include "kernels/prefilter_l1_ring.hpp"
#include "kernels/lvq.hpp"
#include "kernels/lvq_batch.hpp"
#include "kernels/qerror2_lininit.hpp"
