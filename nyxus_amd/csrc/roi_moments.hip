// roi_moments.hip -- ROI contour + 2-D geometric moments (shape: Smoms2D_feature, intensity: Imoms2D_feature),
// SURVEY.md 8(f) #4.
//
//   roi_contour_kernel   /root/reference/src/nyx/features/contour.cpp:381-678  (buildRegularContour + gather_multicontour
//                        + check_loop).  One wave per ROI.  The reference's list algebra (std::list::remove, find_cands over
//                        the unordered list, ...) is O(C^2); here the contour candidates live as bits of a byte plane of the
//                        padded bounding box, so "is my 4- / 8-neighbour still unordered" is one LDS read and the loop
//                        walk is O(C).  The border traces are order-dependent state machines replayed by lane 0 (run starts
//                        come from ballots); the loop walk keeps its state wave-uniform and probes the 8 neighbours with 8 lanes;
//                        plane set-up, the neighbour filter and the X-crossing screen use all lanes.
//                        Output: the merged multicontour in walk order (LR::merge_multicontour), padded coordinates
//                        (the reference adds the bbox origin without removing the one-pixel padding, contour.cpp:673-678).
//   roi_moments_kernel   features/2d_geomoments_basic.cpp:32-376.  One 256-thread workgroup per ROI, four passes over the
//                        cloud (raw, central, weights + weighted raw, weighted central); the distance of a pixel to the contour
//                        is the reference's hill descent over the ORDERED contour (features/pixel.cpp:40-70), and the
//                        weighted intensity passes through float like the reference's vector<float> (pixel.h:8).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "device_math.h"
#include "roi_kernel.h"
#include "launch_util.h"
#include "../../include/nyxhip.h"

namespace nyxhip {

namespace {

constexpr uint8_t kPix = 1, kBorder = 2, kCand = 4, kAlive = 8;

// the value as a VECTOR register the optimiser cannot look through (a store indexed by it takes scalar base + vector offset: no
// 64-bit address arithmetic on the scalar unit, which bounds the contour kernel)
__device__ __forceinline__ int here_i(int v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ int dial_pos(int dx, int dy)     // contour.cpp:218-262
{
    if (dx > 0) return dy < 0 ? 2 : dy > 0 ? -1 : 1;
    if (dx < 0) return dy < 0 ? 4 : dy > 0 ? -3 : 5;
    return dy < 0 ? 3 : dy > 0 ? -2 : 0;
}

} // namespace

// diagnostic builds (-DNYX_CON_EXIT_AT=k): the contour kernel ends after phase k (results are wrong by design)
#ifdef NYX_CON_EXIT_AT
#define CSTAMP(k) do { if ((k) == NYX_CON_EXIT_AT) return; } while (0)
#else
#define CSTAMP(k) do { } while (0)
#endif
template <bool GS>
__global__ __launch_bounds__(64 * kContourWaves) void roi_contour_kernel(const MomArgs A)
{
    // one wave per ROI, kContourWaves ROIs per workgroup (the waves never meet: no workgroup barrier in this kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wslot = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t slot = (uint64_t)blockIdx.x * kContourWaves + (uint64_t)wslot;
    if (slot >= A.grid_rois)
        return;
    uint8_t* const img = GS ? A.sp.scratch + (size_t)slot * A.sp.stride : lds_raw + (size_t)wslot * ((A.plane_cap + 15u) & ~15u);   // [(w+2)*(h+2)] flag plane
    const uint64_t roi = A.sp.roi_index ? A.sp.roi_index[slot] : slot;
    if (roi >= A.n_roi)
        return;
    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    const int w = (int)A.bbox_w[roi], h = (int)A.bbox_h[roi];
    const int W2 = w + 2, H2 = h + 2;
    const uint32_t np = (uint32_t)W2 * (uint32_t)H2;
    if (n == 0 || np > A.plane_cap) {
        if (n != 0 && A.sp.defer_large)
            return;                                   // handled by the spill launch that follows
        if (lane == 0) {
            A.n_contour[roi] = 0;
            if (n != 0) atomicCAS(A.status, 0, NYXHIP_ERR_ROI_TOO_LARGE);
        }
        return;
    }
    uint32_t* const K = A.ws_contour + off;           // contour out: x | y << 16, padded coordinates
    uint32_t* const stk = (uint32_t*)(A.ws_L + off);  // bifurcation stack of the loop walk (the moments kernel reuses the space)

    for (uint32_t i = lane; i < np; i += 64) img[i] = 0;
    wav_sync<GS>();
    for_each_cloud_pixel<64>(A.inten + off, A.x + off, A.y + off, n, lane, [&](uint32_t, uint32_t, uint32_t px, uint32_t py) {   // padded image, contour.cpp:661-666
        if (px < (uint32_t)w && py < (uint32_t)h) img[mad24(py + 1, (uint32_t)W2, px + 1)] = kPix;
    });
    wav_sync<GS>();

    CSTAMP(0);
    // ---- border image: raster scan with the inside / outside state + Moore trace (contour.cpp:395-493) ----------------
    // The scan's `inside` flag is false after every blank position, so it only lives inside a horizontal run of pixels:
    // at the run's first pixel the scan is outside; a border mark switches it inside for the rest of the run, an unmarked
    // pixel starts a trace (which switches it inside when the trace closes).  Run starts are found 64 positions at a
    // time with a ballot; lane 0 replays the state machine on each run in raster order (traces change the marks).
    // The trace itself (:432-470) probes the eight neighbours one at a time, clockwise from `loc`, until it meets a pixel.
    // Here lanes 0-7 read the eight neighbours in that order at once and a ballot gives the number of misses before the hit:
    // one LDS round trip per contour step instead of one per probe.  Direction d = 0..7 is W, NW, N, NE, E, SE, S, SW
    // (offsets -1, -W2-1, -W2, -W2+1, +1, W2+1, W2, W2-1 of the reference's table); after a hit in direction d the search
    // restarts at loc = {7,7,1,1,3,3,5,5}[d].  Nine misses in a row (no neighbour at all) end the trace like `counter2 > 8`.
    {
        constexpr uint32_t kDx = 0x1A90u, kDy = 0xA901u;             // (dx + 1), (dy + 1) of direction d in bits 2d, 2d + 1
        for (uint32_t base = 0; base < np; base += 64) {
            const uint32_t p = base + (uint32_t)lane;
            const bool run_start = p < np && (img[p] & kPix) && !(p > 0 && (img[p - 1] & kPix));
            unsigned long long m = __ballot(run_start);
            while (m) {                                               // wave-uniform from here on
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                bool inside = false;
                for (uint32_t p0 = base + (uint32_t)b; !inside; p0++) {
                    const uint8_t v = img[p0];
                    if (!(v & kPix)) break;                           // the run is over: outside again
                    if (v & kBorder) { inside = true; break; }        // entering an already discovered border
                    if (lane == 0) img[p0] = v | kBorder;             // undiscovered border point: trace around it
                    int pos = (int)p0, loc = 1, counter = 0;
                    for (;;) {
                        // (the kernel is bound by the CU's scalar unit -- 18 k scalar against 8.5 k vector instructions per wave, one
                        //  scalar issue per cycle and CU -- so the step keeps the scalar unit out of what the lanes can do: every lane
                        //  probes (lanes 8 .. 63 repeat lanes 0 .. 7: no lane mask to set up; the position is a pixel, so all eight
                        //  neighbours lie inside the padded plane: no bounds to test), and the winner's position comes out of the
                        //  winning lane's register with one v_readlane instead of a scalar decode of its direction)
                        const int d = (loc - 1 + lane) & 7;
                        const int cpl = pos + mul_i24((int)((kDy >> (2 * d)) & 3u) - 1, W2) + ((int)((kDx >> (2 * d)) & 3u) - 1);   // (24-bit product: full rate)
                        const uint32_t hm = (uint32_t)__ballot((img[cpl] & kPix) != 0) & 0xFFu;
                        if (hm == 0) break;
                        const int kwin = __ffs((int)hm) - 1;
                        const int dk = (loc - 1 + kwin) & 7;
                        const int cp = __builtin_amdgcn_readlane(cpl, kwin);
                        const int nloc = ((dk & ~1) + 7) & 7;
                        if (cp == (int)p0) {
                            counter++;
                            if (nloc == 1 || counter >= 3) { inside = true; break; }
                        }
                        loc = nloc; pos = cp;
                        img[cp] = (uint8_t)(kPix | kBorder);     // (a hit is a pixel and no other flag exists yet: a plain store, no read on the chain; every lane stores the same byte: no lane mask)
                    }
                }
            }
            wav_sync<GS>();
        }
    }
    CSTAMP(1);
    // ---- candidates: border pixels with a border neighbour, bounds as written (:509-552) -----------------------------
    uint32_t n_cand = 0, n_x = 0;
    if (W2 <= 64) {
        // planes up to a wave wide: a row per step, lane = padded column; the border flags of the rows above / below travel in
        // registers and the horizontal neighbours come through DPP lane shifts (one plane read and one plane write per row
        // instead of an integer division and up to eight reads per position).  The reference's bounds are kept as written:
        // the neighbour tests use the UNPADDED sizes on padded coordinates (xx < w - 1, yy < h - 1).
        const bool in_row = lane < W2;
        const uint32_t gx0 = lane > 0 ? 1u : 0u, gx1 = lane < w - 1 ? 1u : 0u;
        auto row_byte = [&](int yy) -> uint32_t { return (in_row && yy < H2) ? (uint32_t)img[(uint32_t)yy * (uint32_t)W2 + (uint32_t)lane] : 0u; };
        uint32_t v_cur = row_byte(0), v_next = row_byte(1);
        uint32_t b_prev = 0, b_cur = (v_cur >> 1) & 1u;                    // kBorder = 2
        uint32_t c_prev2 = 0, c_prev = 0;                                 // candidate flags of rows yy - 2, yy - 1
        for (int yy = 0; yy < H2; yy++) {
            const uint32_t b_next = (v_next >> 1) & 1u;
            const uint32_t gy0 = yy > 0 ? 1u : 0u, gy1 = yy < h - 1 ? 1u : 0u;
            const uint32_t up = gy0 & b_prev, dn = gy1 & b_next;          // (a row guard applies to the whole row: folded into the row's flags)
            const uint32_t has = (gx0 & (lane_minus1(b_cur, 0u) | lane_minus1(up, 0u) | lane_minus1(dn, 0u))) |
                                 (gx1 & (lane_plus1(b_cur, 0u) | lane_plus1(up, 0u) | lane_plus1(dn, 0u))) | up | dn;
            const uint32_t c_cur = b_cur & has;
            if (c_cur) img[(uint32_t)yy * (uint32_t)W2 + (uint32_t)lane] = (uint8_t)(v_cur | kCand | kAlive);
            n_cand += c_cur;
            // X-crossing count (:566-585) of row yy - 1, whose three candidate rows are known now (rows 1 .. H2 - 2 qualify)
            if (yy >= 2)
                n_x += c_prev & c_prev2 & c_cur & lane_minus1(c_prev, 0u) & lane_plus1(c_prev, 0u);
            c_prev2 = c_prev; c_prev = c_cur;
            b_prev = b_cur; b_cur = b_next; v_cur = v_next; v_next = row_byte(yy + 2);
        }
        wav_sync<GS>();
    } else {
    for (uint32_t p = lane; p < np; p += 64) {
        if (!(img[p] & kBorder)) continue;
        const int yy = (int)(p / (uint32_t)W2), xx = (int)(p - (uint32_t)yy * (uint32_t)W2);
        auto bb = [&](int X, int Y) { return (img[(uint32_t)X + (uint32_t)Y * (uint32_t)W2] & kBorder) != 0; };
        bool has = false;
        if (xx > 0) has |= bb(xx - 1, yy);
        if (xx < w - 1) has |= bb(xx + 1, yy);
        if (yy > 0) has |= bb(xx, yy - 1);
        if (yy < h - 1) has |= bb(xx, yy + 1);
        if (xx > 0 && yy > 0) has |= bb(xx - 1, yy - 1);
        if (xx < w - 1 && yy > 0) has |= bb(xx + 1, yy - 1);
        if (xx > 0 && yy < h - 1) has |= bb(xx - 1, yy + 1);
        if (xx < w - 1 && yy < h - 1) has |= bb(xx + 1, yy + 1);
        if (has) { img[p] |= (uint8_t)(kCand | kAlive); n_cand++; }
    }
    wav_sync<GS>();
    CSTAMP(2);
    // ---- X-crossing fix (:566-585): a candidate whose N, S, W, E are all still in the list leaves it; raster order matters
    for (uint32_t p = lane; p < np; p += 64)
        if ((img[p] & kCand) && p >= (uint32_t)W2 && p + (uint32_t)W2 < np && (img[p - W2] & kCand) && (img[p + W2] & kCand) && (img[p - 1] & kCand) &&
            (img[p + 1] & kCand))
            n_x++;
    }
    n_cand = (uint32_t)wave_sum_u64(n_cand);
    n_x = (uint32_t)wave_sum_u64(n_x);
    n_cand = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_cand);
    n_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_x);
    uint32_t n_u = n_cand;
    if (n_x != 0) {
        if (lane == 0)
            for (uint32_t p = (uint32_t)W2; p + (uint32_t)W2 < np; p++)
                if ((img[p] & kAlive) && (img[p - W2] & kAlive) && (img[p + W2] & kAlive) && (img[p - 1] & kAlive) && (img[p + 1] & kAlive)) {
                    img[p] &= (uint8_t)~kAlive;
                    n_u--;
                }
        n_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_u);
        wav_sync<GS>();
    }
    CSTAMP(3);
    // ---- contour by contour (:587-619) with check_loop (:306-379) on the alive bits.  The walk state is wave-uniform;
    //      lanes 0-3 probe the straight neighbours and lanes 4-7 the diagonal ones, each group ordered by falling dial
    //      position (W 5, N 3, E 1, S -2; NW 4, NE 2, SE -1, SW -3), so prune_cands' winner is the lowest set lane.
    // (every lane probes -- lanes 8 .. 63 repeat lanes 0 .. 7 -- and the winner's step comes out of the winning lane's registers with
    //  v_readlane; stores are issued by every lane to the one address: no lane masks, no scalar address arithmetic.  The kernel is
    //  bound by the scalar unit, see the trace above.)
    const int l8 = lane & 7;
    const int pdx = l8 == 0 ? -1 : l8 == 1 ? 0 : l8 == 2 ? 1 : l8 == 3 ? 0 : l8 == 4 ? -1 : l8 == 5 ? 1 : l8 == 6 ? 1 : -1;
    const int pdy = l8 == 0 ? 0 : l8 == 1 ? -1 : l8 == 2 ? 0 : l8 == 3 ? 1 : l8 == 4 ? -1 : l8 == 5 ? -1 : l8 == 6 ? 1 : 1;
    const int noff = pdx + pdy * W2;                              // this lane's neighbour as an offset in the plane
    const int pstep = pdx + pdy * 65536;                          // both steps in one register: x | y << 16 plus this is (x + dx) | (y + dy) << 16 (padded coordinates are >= 1: no borrow leaves a field)
    uint32_t nK = 0, cursor = 0;
    while (n_u != 0) {
        for (;;) {                                                // U.front(): the raster-first unordered pixel
            const uint32_t p = cursor + (uint32_t)lane;
            const unsigned long long m = __ballot(p < np && (img[p] & kAlive));
            if (m) { cursor += (uint32_t)__ffsll((long long)m) - 1; break; }
            cursor += 64;
        }
        const int ox = (int)(cursor % (uint32_t)W2), oy = (int)(cursor / (uint32_t)W2);
        uint32_t ns = 0, nP = 0;
        int looplen = 0, result = -1;
        // (an alive position carries all four flags: clearing kAlive is a plain store of the other three -- no read-modify-write
        //  on the walk's serial chain)
        // the walk's position -- txy = x | y << 16, tpos = y * W2 + x -- and the output index travel in VECTOR registers (the same value
        // in every lane): their updates and the stores' addresses then cost the vector unit one instruction each and the scalar unit none
        uint32_t txy = (uint32_t)here_i((int)((uint32_t)ox | ((uint32_t)oy << 16)));
        int tpos = here_i((int)cursor);
        uint32_t kidx = (uint32_t)here_i((int)nK);
        K[kidx] = txy; img[tpos] = (uint8_t)(kPix | kBorder | kCand);
        kidx++;
        ns = 1; n_u--;
        wav_sync<GS>();
        while (n_u != 0) {
            // (padded coordinates: the walk stands on a pixel, so all eight neighbours lie inside the plane -- the probe is one add
            //  on the position)
            const uint32_t m = (uint32_t)__ballot((img[(uint32_t)(tpos + noff)] & kAlive) != 0) & 0xFFu;
            const uint32_t cands = (m & 0xFu) ? (m & 0xFu) : (m >> 4);   // find_cands :193-216: straight neighbours first
            const bool diag = (m & 0xFu) == 0;
            const int nc = __popc(cands);
            if (nc > 1) { if (lane == 0) stk[nP] = txy; nP++; }
            if (nc == 0) {
                const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)txy);
                const int ddx = (int)(cur & 0xFFFFu) - ox, ddy = (int)(cur >> 16) - oy;
                if (ddx == 1 || ddx == -1 || ddy == 1 || ddy == -1) { looplen++; result = looplen; break; }
                if (nP == 0) { result = 0; break; }
                --nP;
                uint32_t t = lane == 0 ? stk[nP] : 0u;
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                txy = (uint32_t)here_i((int)t); tpos = here_i((int)((t >> 16) * (uint32_t)W2 + (t & 0xFFFFu)));
            } else {
                const int k = (__ffs((int)cands) - 1) + (diag ? 4 : 0);
                looplen++;
                txy += (uint32_t)__builtin_amdgcn_readlane(pstep, k);
                tpos += __builtin_amdgcn_readlane(noff, k);
                K[kidx] = txy;
                img[tpos] = (uint8_t)(kPix | kBorder | kCand);
                kidx++;
                ns++; n_u--;
                wav_sync<GS>();
            }
        }
        if (result < 0) result = looplen;                         // the list ran empty inside the walk
        if (result > 0) nK += ns;                                 // a closed loop joins the multicontour; a failed chain is dropped
    }
    if (lane == 0) A.n_contour[roi] = nK;
}

// ---- moments -----------------------------------------------------------------------------------------------------------
namespace {

constexpr int kMB = 256;

// Sum N doubles across the workgroup in a fixed order; total k lands in dst[k] (LDS), visible to every thread on return.
// (Only the totals' consumers read them: when every thread formed all N totals itself -- 4 N LDS reads, 3 N adds and 2 N
//  registers each -- the 128-register build spilled around every call: 14 GB of scratch writes per 196 k ROIs in the counters.)
template <int N, bool GS>
__device__ __forceinline__ void mom_block_sum(double (&v)[N], double* s_red, double* dst, int tid)
{
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (N > 8) {                                      // transposed wave sums (device_math.h): slot k's total lands in its lane group
        static_assert(N <= 16, "mom_block_sum: at most 16 sums");
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = k < N ? v[k] : 0.0;
        const double tot = wave_transpose_sum16(t, lane);
        if ((lane & 3) == 0 && (lane >> 2) < N) s_red[wave * N + (lane >> 2)] = tot;
    } else if (N > 4) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = k < N ? v[k] : 0.0;
        const double tot = wave_transpose_sum8(t, lane);
        if ((lane & 7) == 0 && (lane >> 3) < N) s_red[wave * N + (lane >> 3)] = tot;
    } else {
#pragma unroll
        for (int k = 0; k < N; k++) v[k] = wave_sum(v[k]);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < N; k++) s_red[wave * N + k] = v[k];
    }
    blk_sync<GS>();
    if (tid < N) dst[tid] = ((s_red[tid] + s_red[N + tid]) + s_red[2 * N + tid]) + s_red[3 * N + tid];
    blk_sync<GS>();
}

// (int)(m / log(m)) for the window widths m the hill descent meets; m <= 10 -> 1 (pixel.cpp:47,66)
__device__ __forceinline__ int descent_step(size_t m, const uint16_t* tab, int tab_n)
{
    if (m <= 10) return 1;
    if ((int)m < tab_n) return (int)tab[m];
    return (int)((double)m / log((double)m));
}

// Pixel2::min_sqdist v2 (pixel.cpp:40-70): hill descent over the ordered contour.  step0 = (int)(n / log(n)).
// SMALL: every coordinate is below 2^15, so the squared distances are exact in 32-bit integers (24-bit multiplies) and the
// whole search runs on integer compares; otherwise the distances are formed in double like the reference's.  The index
// arithmetic is 32-bit either way (a contour has fewer points than the ROI has pixels).
template <bool SMALL>
__device__ __forceinline__ double min_sqdist_v2(int px, int py, const uint32_t* K, int n, int step0, const uint16_t* tab, int tab_n)
{
    if (n == 0) return 0.0;
    using dist_t = typename std::conditional<SMALL, uint32_t, double>::type;
    const uint32_t ppack = ((uint32_t)px & 0xFFFFu) | ((uint32_t)py << 16);
    auto sqd = [&](uint32_t i) -> dist_t {
        const uint32_t k = K[i];
        if (SMALL) {
            // a contour point is x | y << 16 and both coordinates are below 2^15: the difference is one packed 16-bit subtraction,
            // dx^2 + dy^2 one two-element dot product (v_pk_sub_i16 + v_dot2_i32_i16 instead of unpack / subtract / square / add)
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const s16x2 d = __builtin_bit_cast(s16x2, k) - __builtin_bit_cast(s16x2, ppack);
            return (dist_t)(uint32_t)__builtin_amdgcn_sdot2(d, d, 0, false);
        } else {
            const double dx = (double)(int)(k & 0xFFFFu) - (double)px, dy = (double)(int)(k >> 16) - (double)py;
            return (dist_t)(dx * dx + dy * dy);
        }
    };
    dist_t extrem_d = sqd(0);
    if (n == 1) return (double)extrem_d;
    uint32_t a = 0, b = (uint32_t)n, extrem_i = 0;
    uint32_t step = (uint32_t)step0;
    do {
        for (uint32_t i = a + step; i < b; i += step) {
            const dist_t d = sqd(i);
            if (extrem_d > d) { extrem_d = d; extrem_i = i; }
        }
        const uint32_t stepL = extrem_i >= step ? step : extrem_i,
                       stepR = extrem_i + step < (uint32_t)n ? step : (uint32_t)n - extrem_i;
        a = extrem_i - stepL;
        b = extrem_i + stepR;
        step = (uint32_t)descent_step((size_t)(b - a), tab, tab_n);
    } while (b - a > 2);
    return (double)extrem_d;
}

// The same search for boxes whose squared distances stay below 2^17 and contours below 2^15 - 1 points (every compact-staged ROI but
// the 255 .. 256-pixel corner case): distance and index travel as ONE word, d << 15 | (i + 1), the incumbent with a zero index field,
// so a round's "smaller distance wins, the incumbent keeps ties, the first of equal candidates wins" is one v_min_u32 per candidate
// (a compare, a minimum and a select before), and v_dot2_i32_i16 takes a literal zero addend (the two-operand form the compiler picks
// needs a move per candidate).  K1 = the contour array minus one element (candidate i is K1[i + 1]).
__device__ __forceinline__ double min_sqdist_packed(int px, int py, const uint32_t* K1, int n, int step0, const uint16_t* tab, int tab_n)
{
    const uint32_t ppack = ((uint32_t)px & 0xFFFFu) | ((uint32_t)py << 16);
    auto sqd1 = [&](uint32_t ip1) -> uint32_t {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const s16x2 d = __builtin_bit_cast(s16x2, K1[ip1]) - __builtin_bit_cast(s16x2, ppack);
        uint32_t r;
        asm("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(r) : "v"(d));
        return r;
    };
    uint32_t key = sqd1(1u) << 15;
    if (n == 1) return (double)(key >> 15);
    uint32_t a = 0, b = (uint32_t)n, extrem_i = 0;
    uint32_t step = (uint32_t)step0;
    do {
        for (uint32_t ip1 = a + step + 1u; ip1 <= b; ip1 += step) key = min(key, (sqd1(ip1) << 15) | ip1);
        if (key & 0x7FFFu) { extrem_i = (key & 0x7FFFu) - 1u; key &= ~0x7FFFu; }
        const uint32_t stepL = extrem_i >= step ? step : extrem_i,
                       stepR = extrem_i + step < (uint32_t)n ? step : (uint32_t)n - extrem_i;
        a = extrem_i - stepL;
        b = extrem_i + stepR;
        step = (uint32_t)descent_step((size_t)(b - a), tab, tab_n);
    } while (b - a > 2);
    return (double)(key >> 15);
}

__device__ __forceinline__ double ipow(double a, int b) { double r = 1.0; for (int i = 0; i < b; i++) r *= a; return r; }

__device__ void hu7(double _02, double _03, double _11, double _12, double _20, double _21, double _30, double* h)
{   // calcHu_imp, 2d_geomoments_basic.cpp:231-253
    h[0] = _20 + _02;
    h[1] = ipow((_20 - _02), 2) + 4 * (ipow(_11, 2));
    h[2] = ipow((_30 - 3 * _12), 2) + ipow((3 * _21 - _03), 2);
    h[3] = ipow((_30 + _12), 2) + ipow((_21 + _03), 2);
    h[4] = (_30 - 3 * _12) * (_30 + _12) * (ipow(_30 + _12, 2) - 3 * ipow(_21 + _03, 2)) +
           (3 * _21 - _03) * (_21 + _03) * (3 * ipow(_30 + _12, 2) - ipow(_21 + _03, 2));
    h[5] = (_20 - _02) * (ipow(_30 + _12, 2) - ipow(_21 + _03, 2)) + 4 * _11 * (_30 + _12) * (_21 + _03);
    h[6] = (3 * _21 - _03) * (_30 + _12) * (ipow(_30 + _12, 2) - 3 * ipow(_21 + _03, 2)) -
           (_30 - 3 * _12) * (_21 + _03) * (3 * ipow(_30 + _12, 2) - ipow(_21 + _03, 2));
}

} // namespace

// Column layout of one 90-wide block (Feature2D order): RM(13) CM(16) NRM(16) NCM(7) HU(7) WRM(10) WCM(7) WNCM(7) WHU(7)
// diagnostic builds (-DNYX_MOM_EXIT_AT=k): the kernel ends after pass k (results are wrong by design)
#ifdef NYX_MOM_EXIT_AT
#define MSTAMP(k) do { if ((k) == NYX_MOM_EXIT_AT) return; } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif
// OCC: workgroups per CU the register budget is cut for (4: 128 registers, no spills; 6: 80 registers and ~30 spilled words
// outside the pixel loops -- the hill descent is a chain of dependent LDS reads, so when six carve-outs fit a CU the extra
// waves win: 11.7 -> 10.2 ms per 196 k benchmark ROIs).  The launcher picks by the launch's LDS bytes.
// The value again, but opaque to the optimiser at this point (roi_shape.hip: here()): address arithmetic that depends only on the
// thread index is otherwise formed at the kernel's entry and -- alive across every sweep -- spilled there.
__device__ __forceinline__ int mom_here(int v) { asm volatile("" : "+v"(v)); return v; }
// A workgroup-uniform double as a scalar-register pair (the conversions that produce it leave it in vector registers, where it lived --
// and was spilled and reloaded inside the hill descent -- across every sweep)
__device__ __forceinline__ double mom_uniform(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int OCC>
__global__ __launch_bounds__(kMB, OCC) void roi_moments_kernel(const MomArgs A)
{
    __shared__ double s_red[4 * 16];
    __shared__ double s_raw[2][16], s_cen[2][16], s_wraw[2][10], s_wcen[2][7];
    // staged pixels | contour | step table: sized per launch from the batch extrema (launch_moments), so that small ROIs do not
    // pay for the largest ROI the LDS path accepts -- the benchmark ROI needs 25 KB instead of the 38 KB the fixed arrays took
    extern __shared__ __attribute__((aligned(16))) unsigned char mom_lds[];
    uint2* const s_px = (uint2*)mom_lds;                                  // [A.px_cap]
    uint32_t* const s_K = (uint32_t*)(mom_lds + 8u * A.px_cap);           // [A.k_cap]
    uint16_t* const s_step = (uint16_t*)(s_K + A.k_cap);                  // [A.step_cap]
    const int tid = threadIdx.x;
    const uint64_t roi = A.sp.roi_index ? A.sp.roi_index[blockIdx.x] : blockIdx.x;   // (a list: the big boxes of a batch, launch_moments)
    if (roi >= A.n_roi)
        return;
    const uint64_t off = A.px_offset[roi];
    const uint32_t n = (uint32_t)(A.px_offset[roi + 1] - off);
    // (a launch over the bulk of a batch whose big boxes -- planes beyond the LDS contour kernel's -- go through a list of their own)
    if (A.sp.defer_large && (A.bbox_w[roi] + 2u) * (A.bbox_h[roi] + 2u) > A.plane_cap)
        return;
    double* const row_out = A.out + roi * A.ld;
    const bool do_s = (A.mask & NYXHIP_FAM_SMOMS) != 0, do_i = (A.mask & NYXHIP_FAM_IMOMS) != 0;
    if (n == 0) {
        for (int c = tid; c < 90; c += kMB) {
            if (do_s) row_out[A.col_smoms + c] = __longlong_as_double(0x7ff8000000000000LL);
            if (do_i) row_out[A.col_imoms + c] = __longlong_as_double(0x7ff8000000000000LL);
        }
        return;
    }
    const int nK = (int)A.n_contour[roi];
    const bool small_xy = A.bbox_w[roi] + 2u < 32768u && A.bbox_h[roi] + 2u < 32768u;   // integer distances are exact (min_sqdist_v2)
    const uint32_t* K = A.ws_contour + off;
    if (nK <= (int)A.k_cap) {
        for (int i = tid; i < nK; i += kMB) s_K[i] = K[i];
        K = s_K;
    }
    double* const L = A.ws_L + off;

    // window width -> step of the hill descent (first step from n, later ones from windows of at most two steps)
    const int step0 = __builtin_amdgcn_readfirstlane(nK >= 2 ? (int)((double)nK / log((double)nK)) : 1);
    const int tab_n = min((int)A.step_cap, 2 * step0 + 2);
    for (int m = 11 + tid; m < tab_n; m += kMB) s_step[m] = (uint16_t)(int)((double)m / log((double)m));

    // The six sweeps below read the ROI's pixels again and again.  From HBM / L2 every sweep is a chain of dependent round
    // trips (four waves per SIMD hide little of it: the sweeps took 3-4 ms each way); ROIs of up to kMomPxLds pixels are
    // therefore staged in LDS once -- x | y << 16 and the intensity, 8 bytes per pixel -- and swept from there.
    const bool staged = n <= A.px_cap;
    // Compact staging (boxes up to 256 x 256 whose squared diagonal stays inside the logarithm table -- every staged ROI of usual
    // data): intensity u32 | x, y as two bytes | a 16-bit slot per pixel for the squared distance to the contour pass 3 finds --
    // the same 8 bytes per pixel as the general staging, and the per-pixel weight never travels to HBM: the two sweeps that
    // need it look the logarithm up again in the (L2-resident) table.  The weights written as doubles and read back twice were
    // 13 of the kernel's 16 GB of traffic per 196 k benchmark ROIs (the reference keeps them as vector<float> realintens, pixel.h:8).
    const uint32_t bw_ = A.bbox_w[roi], bh_ = A.bbox_h[roi];
    const bool compact = staged && small_xy && bw_ <= 256u && bh_ <= 256u && (bw_ + 2u) * (bw_ + 2u) + (bh_ + 2u) * (bh_ + 2u) < A.log_tab_n;
    uint32_t* const s_v = (uint32_t*)mom_lds;                             // [A.px_cap]   (compact staging)
    uint16_t* const s_xy8 = (uint16_t*)(s_v + A.px_cap);                  // [A.px_cap]   x | y << 8
    uint16_t* const s_d2 = s_xy8 + A.px_cap;                              // [A.px_cap]   squared distance to the contour
    if (compact)
        for_each_cloud_pixel<kMB>(A.inten + off, A.x + off, A.y + off, n, tid, [&](uint32_t i, uint32_t vi, uint32_t xi, uint32_t yi) {
            s_v[i] = vi; s_xy8[i] = (uint16_t)(xi | (yi << 8));
        });
    else if (staged)
        for_each_cloud_pixel<kMB>(A.inten + off, A.x + off, A.y + off, n, tid, [&](uint32_t i, uint32_t vi, uint32_t xi, uint32_t yi) {
            s_px[i] = make_uint2(xi | (yi << 16), vi);
        });
    __syncthreads();
    // the thread index, formed where it is used from the wave number (a scalar) and the lane count: as one value alive across the whole
    // kernel it was spilled at the entry and reloaded per phase (with what else sat at the entry: 44 B of scratch per lane, the
    // 2.2 GB of writes per 196 k ROIs the counters saw beside 0.28 GB of results)
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto TID = [&]() -> int {                             // (volatile: the optimiser would otherwise form it once and keep it)
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return (wave_s << 6) + l;
    };
    auto sweep = [&](auto&& body) {                      // body(i, intensity, x, y) for this thread's pixels i = tid, tid + 256, ...
        if (compact) {
            for (uint32_t i = (uint32_t)TID(); i < n; i += kMB) {
                const uint32_t xy = s_xy8[i];
                body(i, s_v[i], xy & 0xFFu, xy >> 8);
            }
        } else if (staged) {
            for (uint32_t i = (uint32_t)TID(); i < n; i += kMB) {
                const uint2 q = s_px[i];
                body(i, q.y, q.x & 0xFFFFu, q.x >> 16);
            }
        } else
            for_each_cloud_pixel<kMB>(A.inten + off, A.x + off, A.y + off, n, TID(), body);
    };
    MSTAMP(0);
    // ---- pass 1 + 2 as ONE sweep: moments about the box centre o = ((w - 1) / 2, (h - 1) / 2), mu_o[p][q] = sum I (x - o_x)^p (y - o_y)^q for
    //      all p, q in 0..3 -- then, by the binomial theorem, the raw moments (origin 0: calcRawMoments :266-281) and the central ones
    //      (origin = the centroid: :152-181, :298-316) as fixed combinations of the sixteen sums:
    //          m[p][q] = sum_{k <= p, l <= q} C(p, k) C(q, l) e_x^(p - k) e_y^(q - l) mu_o[k][l],   e = o - (new origin).
    //      The reference sweeps the pixels once per origin; here one sweep serves both (a third of the kernel's sums less).  Rounding:
    //      |x - o| <= side / 2, so every term of a combination is at most (1 + |e| / (side / 2))^(p + q) times the size of the terms of
    //      the direct sum -- <= 64 x for an origin one half-side away -- and the combination's error stays below 1e-14 of
    //      m00 (side / 2)^(p + q), an order under the floor the parity tests grant a central moment that cancels (tests/parity.py).
    const double obx = mom_uniform(0.5 * (double)(bw_ - 1u)), oby = mom_uniform(0.5 * (double)(bh_ - 1u));
    __shared__ double s_mo[2][16], s_wo[2][10], s_wo16[2][16];
#pragma unroll 1
    for (int var = 0; var < 2; var++) {
        double acc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) acc[k] = 0;
        if (var ? do_i : do_s)
        sweep([&](uint32_t, uint32_t vi, uint32_t xi, uint32_t yi) {
            const double dx = (double)xi - obx, dy = (double)yi - oby;
            const double xp[4] = {1.0, dx, dx * dx, dx * dx * dx}, yp[4] = {1.0, dy, dy * dy, dy * dy * dy};
            if (var) {
                const double I = (double)vi;
                const double ix[4] = {I, I * xp[1], I * xp[2], I * xp[3]};
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[p * 4 + q] = q ? __builtin_fma(ix[p], yp[q], acc[p * 4 + q]) : acc[p * 4 + q] + ix[p];
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[p * 4 + q] = (p && q) ? __builtin_fma(xp[p], yp[q], acc[p * 4 + q]) : acc[p * 4 + q] + (p ? xp[p] : yp[q]);
            }
        });
        mom_block_sum<16, false>(acc, s_red, s_mo[var], TID());
    }
    __syncthreads();
    // m[p][q] about the origin o - e from the sums about o (binomial coefficients of orders 0..3)
    // (no local arrays: an array indexed at run time lives in scratch memory)
    auto shifted = [](const double* mo, int p, int q, double ex, double ey) -> double {
        auto binom = [](int n, int k) -> double { return (n == 3 && (k == 1 || k == 2)) ? 3.0 : (n == 2 && k == 1) ? 2.0 : 1.0; };
        auto pw = [](double e, int n) -> double { const double e2 = e * e; return n == 0 ? 1.0 : n == 1 ? e : n == 2 ? e2 : e2 * e; };
        double r = 0.0;
        for (int k = 0; k <= p; k++)
            for (int l = 0; l <= q; l++) r += binom(p, k) * binom(q, l) * pw(ex, p - k) * pw(ey, q - l) * mo[k * 4 + l];
        return r;
    };
    if (TID() < 32) {                                       // raw moments: origin 0, e = o
        const int var = TID() >> 4, pq = TID() & 15;
        s_raw[var][pq] = shifted(s_mo[var], pq >> 2, pq & 3, obx, oby);
    }
    __syncthreads();
    if (TID() < 32) {                                       // central moments: origin (m10 / m00, m01 / m00), each variant its own
        const int var = TID() >> 4, pq = TID() & 15;
        const double cx = s_raw[var][4] / s_raw[var][0], cy = s_raw[var][1] / s_raw[var][0];
        s_cen[var][pq] = shifted(s_mo[var], pq >> 2, pq & 3, obx - cx, oby - cy);
    }
    __syncthreads();
    MSTAMP(1);
    MSTAMP(2);
    // (p, q) of the 10 weighted raw moments and of the 7 (weighted / normalized) central ones
    static constexpr int wr_p[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, wr_q[10] = {0, 1, 2, 3, 0, 1, 2, 0, 1, 0};
    static constexpr int nc_p[7] = {0, 0, 1, 1, 2, 2, 3}, nc_q[7] = {2, 3, 1, 2, 0, 1, 0};
    const bool packed = compact && K == s_K && nK >= 1 && nK <= 32766 && (bw_ + 2u) * (bw_ + 2u) + (bh_ + 2u) * (bh_ + 2u) < (1u << 17);   // (min_sqdist_packed)
    // ---- pass 3: log(distance to contour + eps) per pixel (:32-53) and the weighted raw moments (:283-296); the weighted
    //      intensity passes through float (realintens is a vector<float>)
    {
        double as[10], ai[10];
#pragma unroll
        for (int k = 0; k < 10; k++) { as[k] = 0; ai[k] = 0; }
        sweep([&](uint32_t i, uint32_t vi, uint32_t xi, uint32_t yi) {
            // (a squared distance between integer points is an integer: for the small ones the logarithm comes from a table
            //  built once per context with this very expression -- ~100 vector instructions per pixel less)
            // (the contour normally sits in LDS: passing the array itself -- not a pointer that may also be global -- turns the
            //  descent's loads into ds_read instead of flat loads with 64-bit addresses)
            const double dsq = !small_xy ? min_sqdist_v2<false>((int)xi, (int)yi, K, nK, step0, s_step, tab_n)
                             : packed ? min_sqdist_packed((int)xi, (int)yi, s_K - 1, nK, step0, s_step, tab_n)
                             : K == s_K ? min_sqdist_v2<true>((int)xi, (int)yi, s_K, nK, step0, s_step, tab_n)
                                        : min_sqdist_v2<true>((int)xi, (int)yi, K, nK, step0, s_step, tab_n);
            if (compact) { s_d2[i] = (uint16_t)(uint32_t)dsq; return; }       // (dsq <= the box's squared diagonal < log_tab_n <= 65536)
            const double lg = (small_xy && dsq < (double)A.log_tab_n) ? A.log_tab[(uint32_t)dsq] : log(sqrt(dsq) + 0.001);
            L[i] = lg;
        });
        // (the sums run as a sweep of their own over the weights just written -- every thread reads back its own stores: with
        //  the twenty accumulators live across the hill descent the 80-register build spilled and reloaded them per pixel)
        sweep([&](uint32_t i, uint32_t vi, uint32_t xi, uint32_t yi) {
            const double lg = compact ? A.log_tab[s_d2[i]] : L[i];
            const double X = (double)xi - obx, Y = (double)yi - oby;           // (about the box centre, like the sums of pass 1 + 2)
            const double Ws = (double)(float)(1.0 * lg), Wi = (double)(float)((double)vi * lg);
            // (one power of x at a time, the shape sums before the intensity sums, the scheduler kept from interleaving the groups:
            //  with sx[4], ix[4], xp, yp all live beside the twenty accumulators the 80-register build spilled around this loop)
            const double yp[4] = {1.0, Y, Y * Y, Y * Y * Y};
            double xpw = 1.0;
#pragma unroll
            for (int p = 0; p < 4; p++) {
                if (p) xpw = p == 1 ? X : xpw * X;               // X, X * X, (X * X) * X as before
                {
                    const double sx = p ? Ws * xpw : Ws;
#pragma unroll
                    for (int k = 0; k < 10; k++)
                        if (wr_p[k] == p) as[k] = wr_q[k] ? __builtin_fma(sx, yp[wr_q[k]], as[k]) : as[k] + sx;
                }

                {
                    const double ixv = p ? Wi * xpw : Wi;
#pragma unroll
                    for (int k = 0; k < 10; k++)
                        if (wr_p[k] == p) ai[k] = wr_q[k] ? __builtin_fma(ixv, yp[wr_q[k]], ai[k]) : ai[k] + ixv;
                }

            }
        });
        mom_block_sum<10, false>(as, s_red, s_wo[0], TID());
        mom_block_sum<10, false>(ai, s_red, s_wo[1], TID());
    }
    __syncthreads();
    MSTAMP(3);
    // ---- pass 4 without a sweep: the weighted raw moments (origin 0, :283-296) and the weighted central ones about the weighted
    //      origin (:162-167, :318-327) from the ten weighted sums about the box centre, as above.  (The weighted origin may lie far
    //      outside the box when the weighted mass nearly cancels: then |e| >> side and the shifted sum is dominated by its e^(p + q)
    //      term -- as the reference's direct sum is.)
    if (TID() < 32) s_wo16[TID() >> 4][TID() & 15] = 0.0;         // the ten sums in the [p * 4 + q] layout of `shifted` (the other six: zero)
    __syncthreads();
    if (TID() < 20) { const int var = TID() / 10, k = TID() % 10; s_wo16[var][wr_p[k] * 4 + wr_q[k]] = s_wo[var][k]; }
    __syncthreads();
    if (TID() < 20) {
        const int var = TID() / 10, k = TID() % 10;
        s_wraw[var][k] = shifted(s_wo16[var], wr_p[k], wr_q[k], obx, oby);
    }
    __syncthreads();
    if (TID() < 14) {
        const int var = TID() / 7, k = TID() % 7;
        const double ox = s_wraw[var][4] / s_wraw[var][0], oy = s_wraw[var][1] / s_wraw[var][0];
        s_wcen[var][k] = shifted(s_wo16[var], nc_p[k], nc_q[k], obx - ox, oby - oy);
    }
    __syncthreads();
    MSTAMP(4);
    // ---- derived values and output ------------------------------------------------------------------------------------------
    if (TID() < 2 && (TID() ? do_i : do_s)) {
        const int var = TID();
        double* o = row_out + (var ? A.col_imoms : A.col_smoms);
        const double* raw = s_raw[var];
        const double* cen = s_cen[var];
        const double m00 = raw[0], w00 = s_wraw[var][0];
        // x / pow(m, (p + q) / 2 + 1) for p + q = 0..6: the seven powers are m^(1 + j/2) = m^(1 + j div 2) * sqrt(m)^(j mod 2) -- one
        // square root, a few products and seven divisions per normalising mass, where thirty pow() calls (a couple of hundred
        // instructions each, on two lanes of one wave) were a tenth of the kernel.  Tolerance-class values; 1-2 ulp from pow().
        double rm[7], rw[7];
        {
            const double sm = sqrt(m00), sw = sqrt(w00);
            double pm = m00, pw = w00;
#pragma unroll
            for (int j = 0; j < 7; j += 2) {
                rm[j] = 1.0 / pm; rw[j] = 1.0 / pw;
                if (j + 1 < 7) { rm[j + 1] = 1.0 / (pm * sm); rw[j + 1] = 1.0 / (pw * sw); }
                pm *= m00; pw *= w00;
            }
        }
        for (int k = 0; k < 13; k++) o[k] = raw[k];                                       // RM 00..23, 30
        for (int k = 0; k < 16; k++) o[13 + k] = cen[k];                                          // CM
#pragma unroll
        for (int k = 0; k < 16; k++)                                                               // NRM :204-209
            o[29 + k] = raw[k] * rm[(k >> 2) + (k & 3)];
        double nu[7], wn[7];
#pragma unroll
        for (int k = 0; k < 7; k++) {                                                              // NCM :212-217
            nu[k] = cen[nc_p[k] * 4 + nc_q[k]] * rm[nc_p[k] + nc_q[k]];
            o[45 + k] = nu[k];
        }
        hu7(nu[0], nu[1], nu[2], nu[3], nu[4], nu[5], nu[6], o + 52);
        for (int k = 0; k < 10; k++) o[59 + k] = s_wraw[var][k];                                   // WRM
        for (int k = 0; k < 7; k++) o[69 + k] = s_wcen[var][k];                                    // WCM
#pragma unroll
        for (int k = 0; k < 7; k++) {                                                              // WNCM :220-225
            wn[k] = s_wcen[var][k] * rw[nc_p[k] + nc_q[k]];
            o[76 + k] = wn[k];
        }
        hu7(wn[0], wn[1], wn[2], wn[3], wn[4], wn[5], wn[6], o + 83);
    }
}

int launch_roi_contour(const MomArgs& a, void* stream, uint32_t grid)
{
    static DeviceOnce optin;
    if (int orc = optin.run([]() -> int {
        return (int)hipFuncSetAttribute((const void*)roi_contour_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)roi_features_max_lds());
    }))
        return orc;
    if (grid == 0)
        return 0;
    MomArgs b = a;
    b.grid_rois = grid;
    const uint32_t wgs = (grid + kContourWaves - 1) / kContourWaves;
    if (a.sp.scratch)
        hipLaunchKernelGGL(roi_contour_kernel<true>, dim3(wgs), dim3(64 * kContourWaves), 0, (hipStream_t)stream, b);
    else
        hipLaunchKernelGGL(roi_contour_kernel<false>, dim3(wgs), dim3(64 * kContourWaves), kContourWaves * ((a.plane_cap + 15u) & ~15u), (hipStream_t)stream, b);
    return (int)hipGetLastError();
}

__global__ void moments_logtab_kernel(double* tab, uint32_t n)
{
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n) tab[d] = log(sqrt((double)d) + 0.001);
}

int launch_moments_logtab(double* tab, uint32_t n, void* stream)
{
    hipLaunchKernelGGL(moments_logtab_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tab, n);
    return (int)hipGetLastError();
}

int launch_roi_moments(const MomArgs& a, void* stream, uint32_t grid)
{
    if (grid == 0)
        return 0;
    const uint32_t dyn = 8u * a.px_cap + 4u * a.k_cap + 2u * a.step_cap;
    // static LDS of the kernel (exchange area + the four total blocks) is 1296 B; LDS is allocated in 1280-byte granules
    const uint32_t granules = (dyn + 1296u + 1279u) / 1280u;
    // (A/B knob.  The six-per-CU build carries 12 spilled VGPRs -- 40 B of scratch per lane, the 1.9 GB the counters see written per
    //  196 k ROIs -- and is still the faster one: 16.5 ms for both moment families against 17.7 ms for the spill-free build at four
    //  workgroups per CU and 16.7 ms for a 96-register build, which spills 10 all the same: the kernel wants 108.)
    static const int occ_knob = [] { const char* e = getenv("NYXHIP_MOM_OCC"); return e && *e ? atoi(e) : 6; }();
    if (occ_knob >= 6 && 6u * granules * 1280u <= (uint32_t)roi_features_max_lds())
        hipLaunchKernelGGL(roi_moments_kernel<6>, dim3(grid), dim3(kMB), dyn, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(roi_moments_kernel<4>, dim3(grid), dim3(kMB), dyn, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

} // namespace nyxhip
