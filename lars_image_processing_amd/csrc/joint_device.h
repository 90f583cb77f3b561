// Pieces shared by the one-read statistics kernels: the two-reader counting kernel and the finish kernel (joint.hip)
// and the one-reader counting kernel over windowed tables with its window prediction (joint_win.hip).
#pragma once
#include <string.h>

#include <type_traits>

#include "v2_device.h"

namespace lars {

#define JH_K 5u                                          /* m = n + JH_K * x: odd, and 5 x 5 neighbourhoods tile 25 consecutive values */
#define JH_DWORDS 32768
#define JH_THREADS 1024
#define JH_PERIOD_STEPS 12                    /* steps of 4096 pixels between two scans */
#define JH_PROMOTE_MASK 0x0000C000u           /* a low half >= 16384 */
#define JH_LIST_CAP 1024
#define JH_MAX_WG_PIXELS (1ll << 24)

struct JointWin;
struct JointCountParams {
    const uint8_t *tiles;
    long long npix;
    long long ntiles;
    long long chunk_quads;                    // quads per chunk: a multiple of 1024 (the last chunk of a tile takes the rest)
    unsigned int *part;                       // [ntiles][S][K][32768][2] uint32: counts of the cells (D, 0), (D, 1)
    unsigned int *error;                      // set to 1 if a list overflows (cannot happen: see JH_MAX_WG_PIXELS)
    int K;                                    // chunks per tile
    int S;                                    // streams counted: 1 or 2
    unsigned int streams;                     // bit 0: (n, r) pairs, bit 1: (n, g) pairs
    struct JointWin *win;                     // [ntiles] or null: which tiles are counted on windowed tables (joint_win.hip)
    int pass;                                 // 0: the first count (k_joint_count skips the windowed tiles); 1: the recount of the tiles whose window missed
    uint2 *rows;                              // [ntiles][S][K]: (first, one past the last) block of 8 table rows that holds a count -- what k_joint_finish walks
    uint2 *list;                              // k_joint_count_win: [ntiles * K][JW_LIST_CAP] moved dwords (dword, value)
};

__device__ inline void jh_add(unsigned int addr, unsigned int val, char *tab)
{
    __hip_atomic_fetch_add(reinterpret_cast<unsigned int *>(tab + addr), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The hand-over of a moved dword at publish time: an add to a word of the workgroup's OWN published table, after a barrier behind the
// stores that wrote it.  Workgroup scope: stores and adds meet in this XCD's L2; the next kernel sees them through the launch boundary.
__device__ inline void jh_publish_add(unsigned int *word, unsigned int val)
{
    __hip_atomic_fetch_add(word, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// nn + 5 px in one full-rate instruction (px < 2^24; the compiler's own choice for * 5 + is the quarter-rate v_mad_u64_u32)
__device__ inline unsigned int jh_mad5(unsigned int px, unsigned int nn)
{
    unsigned int r;
    asm("v_mad_u32_u24 %0, %1, 5, %2" : "=v"(r) : "v"(px), "v"(nn));
    return r;
}
static_assert(JH_K == 5u, "jh_mad5 spells the factor out");

// Runs of equal pixels (nodata borders, saturated sky, flat fills): when at least half of a wave's lanes hold the same quad as the lane before
// them, the lanes that START a run add the whole run's count and the others add nothing -- 64 lanes queueing on a few LDS words become
// a few lanes.  jh_run_head: does this lane start a run (its quad differs from the previous lane's; rows of 16 lanes start one anyway)?
// The cheap half of the test (first dword only) runs for every quad: one v_mov_b32_dpp + one compare.
__device__ inline unsigned int jh_prev_lane(unsigned int w)
{
    return (unsigned int)__builtin_amdgcn_update_dpp((int)~w, (int)w, 0x111 /* row_shr:1 */, 0xF, 0xF, false);   // first lane of a row: ~w
}
// (The gate's own compare lets the first lane of each row of 16 read 0 from beyond the row -- one instruction instead of three; those four
// lanes may count as repeats when their dword is 0, which moves a threshold of 32 lanes by at most 4.  The run path itself decides heads
// with jh_prev_lane.)  The count and its compare stay on the scalar unit.
__device__ inline bool jh_mostly_runs(unsigned int w0)
{
    const unsigned int prev = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)w0, 0x111 /* row_shr:1 */, 0xF, 0xF, true);
    const unsigned long long same = __builtin_amdgcn_ballot_w64(prev == w0);
    unsigned int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(same) : "scc");
    return n >= 32u;
}
// length of the run a head lane starts: up to the next head, the first inactive lane, or the wave's end
__device__ inline unsigned int jh_run_length(bool head, int lane)
{
    const unsigned long long stops = __builtin_amdgcn_ballot_w64(head) | ~__builtin_amdgcn_ballot_w64(true);
    const unsigned long long above = (stops >> lane) >> 1;
    return above ? (unsigned int)__builtin_ctzll(above) + 1u : 64u - (unsigned int)lane;
}

// (D, h) of a pair of samples, and back
__device__ inline unsigned int jh_m(unsigned int n, unsigned int x) { return (n + JH_K * x) & 255u; }
__device__ inline unsigned int jh_n_of(unsigned int D, unsigned int h)
{
    const unsigned int x = D >> 7, m = (D & 127u) | (h << 7);
    return (m - JH_K * x) & 255u;
}


// ---- windowed tables (joint_win.hip) ------------------------------------------------------------------------------
// After the percentile white balance every sample at or below p2 is level 0 and every sample at or above p98 is level 255
// (process-images.py:438: clip((ch - p2) / (p98 - p2) * 255, 0, 255)), so for the statistics a pair table only has to tell the
// samples INSIDE a window [lo, hi] with lo < p2 and p98 < hi apart: whatever lies below counts as lo, whatever lies above as hi.
// Windows on red and green (NIR keeps its 256 values) shrink the two pair tables of a tile to (nr + ng) x 256 cells of 16 bits --
// both fit ONE workgroup's LDS when nr + ng <= JW_MAX_ROWS, and the tile is then read by one CU instead of two.
// The windows come from a subsample (k_joint_predict); the exact percentiles follow from the counted marginals as before and
// k_joint_finish checks that they lie strictly inside the window -- a tile whose window missed is counted again on full tables.
#define JW_PITCH 133u                          /* dwords per table row: 128 + 5, so that the LDS bank is (n + 5 x') mod 32 as in the full tables */
#define JW_MAX_ROWS 306                        /* nr + ng: 306 x 133 dwords = 162792 bytes of the CU's 163840 */
#define JW_TAB_DWORDS 40700                    /* JW_MAX_ROWS x JW_PITCH, rounded up to whole uint4 */
// The windowed kernel sweeps its 16-bit counters every 15 steps of 4096 pixels and moves a dword onto the list from 4096 on: 4095 + 15 x 4096
// = 65535.  (The full-table kernel: every 12 steps, from 16384 on.)  A fifth fewer sweeps -- each reads the tables in use from the LDS, which
// is this kernel's busiest unit (profiles/r05_joint_window_scan_period.txt) -- for a longer list: two tables x 2^24 pixels / 4096.
#define JW_PERIOD_STEPS 15
#define JW_PROMOTE_MASK 0x0000F000u            /* a low half >= 4096 */
#define JW_LIST_CAP 8192                       /* moved dwords of a workgroup, in global memory */
#define JW_MIN_PIXELS (1ll << 20)              /* smaller tiles are not worth a window (zeroing + publishing the tables) */

struct JointWin {                              // per tile, in the scratch of lars_d_stats_joint
    unsigned int mode;                         // 0: full tables, two readers (k_joint_count); 1: windows on red and green, NIR whole (k_joint_count_win<.., false>);
                                               // 2: windows on all three channels (k_joint_count_win<.., true>)
    unsigned short lo_r, nr;                   // red window: samples lo_r .. lo_r + nr - 1 (rows 0 .. nr - 1 of table A)
    unsigned short lo_g, ng;                   // green window (rows nr .. nr + ng - 1)
    unsigned int flag;                         // set by k_joint_finish: an order statistic of np.percentile fell onto a window's edge
    unsigned short lo_n, nn;                   // NIR window (mode 2; otherwise 0, 256)
    unsigned short pitch, half;                // dwords per table row; cells n' < half in a dword's low part, n' - half in its high half (mode 1: 133, 128)
};
static_assert(sizeof(JointWin) == 24, "JointWin: six dwords");

// Three windows (mode 2).  NIR clamps like the other two -- the white balance maps it through its own percentiles -- so a row needs
// only nn cells: half = ceil(nn / 2) dwords, padded to a pitch whose multiples spread a 5 x 5 neighbourhood of (row, cell) pairs over
// 25 banks (pitch mod 32 in {5, 6, 26, 27}).  Three windows of up to about 196 values share one workgroup's LDS this way (two of 153 with
// NIR whole): photographs whose histograms span 60-77 % of the 8-bit range.  The kernel pays three more packed instructions per pixel
// pair for it (n' -> dword and half by compare instead of by bit 7) and runs 1-4 % behind the NIR-whole form, 17 % ahead of two readers.
__host__ __device__ inline unsigned int jw_pitch_for(unsigned int half)
{
    unsigned int p = half;
    for (;; ++p) {
        const unsigned int r = p & 31u;
        if (r == 5u || r == 6u || r == 26u || r == 27u) return p;
    }
}

// joint_win.hip
void joint_predict_launch(const uint8_t *tiles, long long ntiles, long long npix, int channels, JointWin *win, int test_mode, hipStream_t s);
void joint_count_win_launch(const JointCountParams &C, int channels, int depth, hipStream_t s);

}  // namespace lars
