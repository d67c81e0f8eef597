// Streaming brick kernel of the unprojection: C = 32 channels, 1..16 views, grid dimensions multiples of 8.
// Included by unproject.hip (inside namespace mvx, after the helpers it uses).
//
// Like unproject_brick32, a workgroup of 512 threads owns a brick of 8 x 8 x 8 voxels (one voxel per lane) and reads
// the bilinear taps of a view from a patch of that view's map staged in LDS.  Unlike it, a workgroup is resident
// for the whole launch: it walks its share of the bricks and treats (brick, view) as one unit of a pipeline with two
// patch slots.  While the taps of unit p are read from slot p & 1, the patch of unit p + 1 -- the next view of the
// brick, or the first view of the workgroup's next brick -- is on its way into the other slot (global_load_lds),
// so no wave waits for a copy it has just issued, a brick's output stores drain while the next brick is sampled,
// and no workgroup slot stands empty between bricks.  One barrier per unit: a wave waits for the copies it issued
// one unit ago (everything younger in its vector-memory queue is the 8 stores of a brick's last unit, hence the
// counted s_waitcnt), then the barrier tells it that everybody's copies have landed and that everybody has finished
// reading the slot that is about to be refilled.
//
// Views are aggregated one at a time in registers, for any number of views.  sum / conf / max run the same
// operations in the same order as the tile kernel (bit-identical).  The view softmax sum_v x_v e^{x_v} / sum_v e^{x_v}
// is accumulated without subtracting the maximum over views -- num += 2^{x'} x', den += 2^{x'} with x' = x log2 e,
// two registers per channel however many views there are -- which is the same quotient as long as no term
// overflows and the largest term is not flushed.  That is checked, not assumed: every denominator must lie in
// [2^-80, 2^100] (terms below 2^-46 of the largest one may then have been lost to underflow at most); a brick that
// fails the check is computed again by the direct path, which evaluates the shifted form.  Results differ from the
// shifted form by the rounding of the exponentials only (a few float32 ulps).
//
// The patch rectangle of every view of a brick comes from the projections of the brick's 8 corner voxels (exact
// for voxel positions that are affine in the index and in front of the camera), computed two bricks ahead by one
// wave; every lane still checks its own taps against the rectangle.  A brick with a patch over 16 x 16 pixels, a
// corner behind a camera, a lane whose taps fall outside, or a failed softmax check is put on a list and computed
// after the workgroup's last unit by the direct path: taps from global memory with the tile kernel's access shape
// (8 lanes x 16 bytes per voxel) and the tile kernel's arithmetic, through per-wave tables in the patch slots.
// Such bricks cluster in space (near a camera), so a workgroup's share rotates through the brick positions: at step
// j the workgroups of an XCD work on one run of consecutive bricks (neighbours share their patches in its L2), and
// which of them a workgroup takes shifts by a fixed odd amount from step to step.
#pragma once

#ifdef MVX_STREAM_DEBUG
// Development builds (tools/brick_lab.hip) only: every global access of the kernel is checked against the buffers
// it may touch; an access outside is recorded and redirected to the start of the buffer instead of being made.
// The record travels in the (otherwise unused) coordinate-volume pointer of a grid-parameter launch.
struct StreamDebug {
    const char* lo[2];   // feature maps, output volumes
    const char* hi[2];
    long long n_frames;
    unsigned mask;                    // 1: patch copies, 2: sampling and stores, 4: deferred bricks
    unsigned long long count[4];      // feature reads, volume writes, frame index, table entries
    unsigned long long first[4][4];   // value, block, thread, line
};
#define MVX_DBG (reinterpret_cast<StreamDebug*>(const_cast<float*>(coord.volume)))
__device__ __forceinline__ void stream_debug_note(StreamDebug* d, int kind, long long value, int line) {
    if (atomicAdd(&d->count[kind], 1ull) == 0) {
        d->first[kind][0] = static_cast<unsigned long long>(value);
        d->first[kind][1] = blockIdx.x;
        d->first[kind][2] = threadIdx.x;
        d->first[kind][3] = static_cast<unsigned long long>(line);
    }
}
// pointer into buffer `kind`, redirected to its start when [ptr, ptr + bytes) leaves it
template <class P>
__device__ __forceinline__ P stream_debug_ptr(StreamDebug* d, int kind, P p, int bytes, int line) {
    const char* c = (const char*)(p);
    if (c >= d->lo[kind] && c + bytes <= d->hi[kind]) return p;
    stream_debug_note(d, kind, c - d->lo[kind], line);
    return (P)(d->lo[kind]);
}
#define MVX_DBG_NOTE(kind, value) stream_debug_note(MVX_DBG, kind, static_cast<long long>(value), __LINE__)
#define MVX_DBG_ON(bit) ((MVX_DBG->mask & (bit)) != 0)
#define MVX_DBG_PTR(kind, ptr, bytes) stream_debug_ptr(MVX_DBG, kind, (ptr), (bytes), __LINE__)
#else
#define MVX_DBG_PTR(kind, ptr, bytes) (ptr)
#define MVX_DBG_ON(bit) true
#endif

constexpr int kStreamMaxViews = 16;
constexpr int kStreamMaxBricks = 1024;  // bricks per workgroup and launch (the host splits larger batches)
constexpr unsigned kStreamRotate = 27;  // odd

struct StreamRect {
    int ox, oy, rows, fits;  // patch origin (pixels), rows staged, 1 if the patch fits a slot
};
struct StreamShared {
    // entry = brick ordinal & 3: bricks j - 1 (softmax re-check), j, j + 1 (staged ahead) and j + 2 (being computed)
    double cam[4][kStreamMaxViews * 12];
    StreamRect rect[4][kStreamMaxViews];
    int fits[4];  // every view of the brick fits
    int bad[4];   // some lane's taps lie outside a staged patch
    int redo[4];  // the softmax range check failed
    unsigned short deferred[kStreamMaxBricks];  // brick ordinals left to the direct path
};

struct BrickItem {
    int64_t b;  // frame of the launch
    int bx, by, bz;
};
__device__ __forceinline__ BrickItem decode_brick(unsigned work, FastDiv bricks_per_frame, FastDiv nbz, FastDiv nby, int vx) {
    BrickItem it;
    const unsigned frame = fast_div(work, bricks_per_frame);
    it.b = frame;
    const unsigned brick = work - frame * bricks_per_frame.divisor;
    const unsigned brick_xy = fast_div(brick, nbz);
    it.bz = static_cast<int>(brick - brick_xy * nbz.divisor);
    it.bx = static_cast<int>(fast_div(brick_xy, nby));
    it.by = static_cast<int>(brick_xy - static_cast<unsigned>(it.bx) * nby.divisor);
    if ((nby.divisor & 3u) == 0 && ((vx / kBrick) & 1) == 0) {
        // columns of bricks (all z) in tiles of 2 (x) by 4 (y), as in unproject_brick32
        const unsigned t = brick_xy >> 3, in = brick_xy & 7u, tiles_y = nby.divisor >> 2;
        const unsigned tx = t / tiles_y, ty = t - tx * tiles_y;
        it.bx = static_cast<int>(tx * 2 + (in >> 2));
        it.by = static_cast<int>(ty * 4 + (in & 3u));
    }
    return it;
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// s_waitcnt lgkmcnt(0) + s_barrier without the vmcnt(0) that __syncthreads() adds while LDS copies are in flight
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int AGG, bool GRID>
__global__ __launch_bounds__(kBrickThreads, 4) void unproject_brick_stream(
    const float* __restrict__ feat_cl, const float* __restrict__ proj, VoxelSource coord, const float* __restrict__ vconf,
    int V, int H, int W, int vx, int vy, int vz, FastDiv bricks_per_frame, FastDiv nbz, FastDiv nby, unsigned n_items,
    double sx, double sy, float* __restrict__ vol) {
    constexpr int C = 32;
    extern __shared__ __align__(128) unsigned char smem[];
    StreamShared* sh = reinterpret_cast<StreamShared*>(smem + 2 * kPatchBytes);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6);
    // Step j: the gridDim.x workgroups take the items j gridDim.x ... (j + 1) gridDim.x - 1 (virtual block ids: the
    // blocks of one XCD get consecutive items, see xcd_contiguous); within its XCD's run a workgroup's place rotates.
    const unsigned G = gridDim.x, Gx = G >> 3, my_xcd = blockIdx.x & 7u, my_r = blockIdx.x >> 3;
    const bool rotate = (G & 7u) == 0 && (Gx & (Gx - 1)) == 0;  // a power of two: any odd shift visits every place
    auto virtual_id = [&](unsigned j) -> unsigned {
        return rotate ? (((Gx * j + ((my_r + j * kStreamRotate) & (Gx - 1))) << 3) | my_xcd) : blockIdx.x + j * G;
    };
    const unsigned full_steps = n_items / G;
    const unsigned n_my = full_steps + ((full_steps * G < n_items && virtual_id(full_steps) < n_items) ? 1u : 0u);
    // lane -> voxel of the brick (as in unproject_brick32: the 16 lanes one ds_read_b128 cycle serves lie on 2 (y) x 8 (z))
    const int chunk = (lane >> 2) & 7;
    const int grp = ((lane >> 5) << 1) | (__builtin_popcount(chunk) & 1);
    const int idx = ((chunk >> 1) << 2) | (lane & 3);
    const int lx = wave, ly = (grp << 1) | (idx >> 3), lz = idx & 7;
    const int64_t n_vox = static_cast<int64_t>(vx) * vy * vz;
    const size_t img_stride = static_cast<size_t>(H) * W * C;
    const float wscale = AGG == MVX_AGG_SOFTMAX ? kLog2e : 1.0f;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_byte*)(smem)));

    auto item_of = [&](unsigned j) {
        BrickItem it = decode_brick(xcd_contiguous(virtual_id(j), n_items), bricks_per_frame, nbz, nby, vx);
#ifdef MVX_STREAM_DEBUG
        if (it.b < 0 || it.b >= MVX_DBG->n_frames || it.bx < 0 || it.bx * kBrick >= vx || it.by < 0 || it.by * kBrick >= vy ||
            it.bz < 0 || it.bz * kBrick >= vz) {
            MVX_DBG_NOTE(2, it.b);
            it.b = 0;
            it.bx = it.by = it.bz = 0;
        }
#endif
        return it;
    };
    auto position = [&](const BrickItem& it, int i0, int i1, int i2, double& X, double& Y, double& Z) {
        if (GRID) {
            FrameGrid fg;
            fg.load(coord.grid, coord.first_frame + it.b, vy);
            float x, y, z;
            fg.at(i0, i1, i2, x, y, z);
            X = x;
            Y = y;
            Z = z;
        } else {
            const int64_t n = (static_cast<int64_t>(i0) * vy + i1) * vz + i2;
            const float* cp = coord.volume + (it.b * n_vox + n) * 3;
            X = cp[0];
            Y = cp[1];
            Z = cp[2];
        }
    };

    // One wave: cameras (float64, scaled) and patch rectangles of brick ordinal j into entry j & 3.
    auto compute_rect = [&](unsigned j) {
        const unsigned e3 = j & 3u;
        const BrickItem it = item_of(j);
        double* cam = sh->cam[e3];
        for (int e = lane; e < V * 12; e += 64) cam[e] = camera_entry(proj + (it.b * V + e / 12) * 12, e % 12, sx, sy);
        const int corner = lane & 7;
        double X, Y, Z;
        position(it, it.bx * kBrick + ((corner & 4) ? kBrick - 1 : 0), it.by * kBrick + ((corner & 2) ? kBrick - 1 : 0),
                 it.bz * kBrick + ((corner & 1) ? kBrick - 1 : 0), X, Y, Z);
        bool all_fit = true;
        for (int v0 = 0; v0 < V; v0 += 8) {
            const bool live = v0 + (lane >> 3) < V;
            const int view = live ? v0 + (lane >> 3) : V - 1;
            const TapXY t = project_tap_xy(cam + view * 12, X, Y, Z, H, W, 1.0f);
            int mnx = t.behind ? INT_MAX : t.x0, mny = t.y0, mxx = mnx, mxy = mny;
            mnx = min(mnx, dpp_movi<0xB1>(mnx));
            mxx = max(mxx, dpp_movi<0xB1>(mxx));
            mny = min(mny, dpp_movi<0xB1>(mny));
            mxy = max(mxy, dpp_movi<0xB1>(mxy));
            mnx = min(mnx, dpp_movi<0x4E>(mnx));
            mxx = max(mxx, dpp_movi<0x4E>(mxx));
            mny = min(mny, dpp_movi<0x4E>(mny));
            mxy = max(mxy, dpp_movi<0x4E>(mxy));
            mnx = min(mnx, dpp_movi<0x141>(mnx));
            mxx = max(mxx, dpp_movi<0x141>(mxx));
            mny = min(mny, dpp_movi<0x141>(mny));
            mxy = max(mxy, dpp_movi<0x141>(mxy));
            StreamRect r;
            r.ox = min(max(mnx, 0), W - 2);
            r.oy = min(max(mny, 0), H - 2);
            const int ex = min(max(mxx, 0), W - 2) + 1, ey = min(max(mxy, 0), H - 2) + 1;  // last column / row read
            r.rows = ey - r.oy + 1;
            // a corner behind the camera reads as INT_MAX: no patch
            r.fits = (mxx != INT_MAX) && (ex - r.ox + 1 <= kPatchW) && (r.rows <= kPatchRows);
            if (corner == 0 && live) sh->rect[e3][view] = r;
            all_fit = all_fit && (__ballot(live && !r.fits) == 0ull);
        }
        if (lane == 0) {
            sh->fits[e3] = all_fit ? 1 : 0;
            sh->bad[e3] = 0;
            sh->redo[e3] = 0;
        }
    };

    // Starts the copy of the patch of (brick item `it`, view v, rectangle in entry e3) into slot `slot`: rows of 16
    // pixels, half a row (8 pixels x 128 bytes) per wave-instruction, row py by wave 2 (py & 3) + half.  The 16-byte
    // channel quad q of a pixel in patch row py sits at position q ^ (py & 7) of the pixel's 128 bytes.
    auto stage = [&](const BrickItem& it, int v, unsigned e3, int slot) {
        const int ox = uniform(sh->rect[e3][v].ox), oy = uniform(sh->rect[e3][v].oy), rows = uniform(sh->rect[e3][v].rows);
        global_byte* img = (global_byte*)(feat_cl + (it.b * V + v) * img_stride);
        lds_byte* dst = (lds_byte*)(smem) + slot * kPatchBytes;
        if (!MVX_DBG_ON(1u)) return;
        const int half = wave & 1, prow = wave >> 1;
        const unsigned col = static_cast<unsigned>(min(ox + (half << 3) + (lane >> 3), W - 1)) * kPixelBytes;
        const unsigned sw = static_cast<unsigned>((lane & 7) ^ prow) << 4;  // rows py = prow, prow + 8: py & 7 = prow
        const unsigned voff0 = col + sw, voff1 = col + (sw ^ 64u);           // rows prow + 4, prow + 12: py & 7 = prow ^ 4
#pragma unroll
        for (int k = 0; k < kPatchRows / 4; ++k) {
            const int py = k * 4 + prow;  // wave-uniform
            if (py < rows) {
                const unsigned row_off = static_cast<unsigned>((oy + py) * W) * kPixelBytes;  // scalar
                __builtin_amdgcn_global_load_lds(MVX_DBG_PTR(0, img + row_off + ((k & 1) ? voff1 : voff0), 16),
                                                 dst + (py * kPatchW + (half << 3)) * kPixelBytes, 16, 0, 0);
            }
        }
    };

    // Direct path for one brick (ordinal j), cameras in sh->cam[0]: per-wave tables of 8 KB in the patch slots.
    auto direct = [&](unsigned j) {
        const BrickItem it = item_of(j);
        const int64_t n = (static_cast<int64_t>(it.bx * kBrick + lx) * vy + it.by * kBrick + ly) * vz + it.bz * kBrick + lz;
        double X, Y, Z;
        position(it, it.bx * kBrick + lx, it.by * kBrick + ly, it.bz * kBrick + lz, X, Y, Z);
        const float* fb = feat_cl + it.b * V * img_stride;
        const int per_round = V <= 4 ? 64 : (V <= 8 ? 32 : 16);  // V * per_round * 20 + per_round * 4 <= 8192 bytes
        unsigned char* mine = smem + wave * 8192;
        float4* wts = reinterpret_cast<float4*>(mine);                                        // [V][per_round]
        unsigned* pix = reinterpret_cast<unsigned*>(mine + V * per_round * sizeof(float4));   // [V][per_round]
        unsigned* vox = pix + V * per_round;                                                  // [per_round]
        const int myslot = ly * kBrick + lz;  // consecutive along z
        const int q = lane & 7;
        for (int r0 = 0; r0 < 64; r0 += per_round) {
            const bool writes = myslot >= r0 && myslot < r0 + per_round;
            for (int v = 0; v < V; ++v) {
                const TapXY t = project_tap_xy(sh->cam[0] + v * 12, X, Y, Z, H, W, wscale);
                if (writes) {
                    wts[v * per_round + myslot - r0] = make_float4(t.w00, t.w01, t.w10, t.w11);
                    pix[v * per_round + myslot - r0] = static_cast<unsigned>(t.yb * W + t.xb);
                }
            }
            if (writes) vox[myslot - r0] = static_cast<unsigned>(n);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the tables are this wave's own: no barrier
            for (int s0 = 0; s0 < per_round; s0 += 8) {
                const int t = s0 + (lane >> 3);
#ifdef MVX_STREAM_DEBUG
                {
                    bool ok = vox[t] < static_cast<unsigned>(n_vox);
                    for (int v = 0; v < V; ++v) ok = ok && pix[v * per_round + t] <= static_cast<unsigned>((H - 2) * W + W - 2);
                    if (!ok) {
                        MVX_DBG_NOTE(3, vox[t]);
                        continue;
                    }
                }
#endif
                float* dst = vol + it.b * C * n_vox + vox[t];
                if (V == 4) {
                    float4 w4[4];
                    Taps4 f[4];
                    float cc[4][4] = {};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        w4[v] = wts[v * per_round + t];
                        f[v] = load_taps(fb + v * img_stride, pix[v * per_round + t], W, C, q);
                        if (AGG == MVX_AGG_CONF) {
                            const float4 c4 = *reinterpret_cast<const float4*>(vconf + (it.b * 4 + v) * C + q * 4);
                            cc[v][0] = c4.x;
                            cc[v][1] = c4.y;
                            cc[v][2] = c4.z;
                            cc[v][3] = c4.w;
                        }
                    }
                    float val[4][4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) interpolate(f[v], w4[v], val[v]);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        dst[static_cast<int64_t>(q * 4 + k) * n_vox] = combine_four<AGG>(
                            val[0][k], val[1][k], val[2][k], val[3][k], cc[0][k], cc[1][k], cc[2][k], cc[3][k]);
                } else {
                    Aggregator<AGG> agg;
                    agg.init();
                    for (int v = 0; v < V; ++v) {
                        const float4 w4 = wts[v * per_round + t];
                        Tap tap;
                        tap.w00 = w4.x;
                        tap.w01 = w4.y;
                        tap.w10 = w4.z;
                        tap.w11 = w4.w;
                        tap.pixel = pix[v * per_round + t];
                        float val[4], conf[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        sample4(fb + v * img_stride, tap, W, C, q, val);
                        if (AGG == MVX_AGG_CONF) {
                            const float4 c4 = *reinterpret_cast<const float4*>(vconf + (it.b * V + v) * C + q * 4);
                            conf[0] = c4.x;
                            conf[1] = c4.y;
                            conf[2] = c4.z;
                            conf[3] = c4.w;
                        }
                        agg.add(val, conf);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) dst[static_cast<int64_t>(q * 4 + k) * n_vox] = agg.result(k);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // table reads done before the next round overwrites them
        }
    };

    // ---- prologue: rectangles of the first two bricks
    if (wave < 2 && static_cast<unsigned>(wave) < n_my) compute_rect(static_cast<unsigned>(wave));
    lds_barrier();

    unsigned p = 0;               // patch units so far: a unit's patch lives in slot p & 1
    bool primed = false;          // the first patch of the brick about to start is already on its way (into slot p & 1)
    bool stores_behind = false;   // this wave's last 8 vector-memory operations are a brick's output stores
    unsigned n_deferred = 0;      // bricks left to the direct path (the same count in every wave)
    auto defer = [&](unsigned j) {
        if (tid == 0) sh->deferred[n_deferred] = static_cast<unsigned short>(j);
        n_deferred += 1;
    };
    bool check_prev = false;      // brick j - 1 went through the pipeline: its softmax flag is read at the next barrier

    for (unsigned j = 0; j < n_my; ++j) {
        const unsigned e3 = j & 3u;
        const bool rect_ahead = j + 2 < n_my && wave == static_cast<int>(j & 7u);
        // Table entry j was written during brick j - 2 and every brick has at least one barrier after the point where
        // it writes an entry, so the entry is complete here.
        if (uniform(sh->fits[e3]) == 0) {
            // a patch does not fit.  (Every wave here has passed the last barrier of brick j - 1, so every wave has
            // left brick j - 2, whose table entry compute_rect(j + 2) takes over.)
            if (AGG == MVX_AGG_SOFTMAX && check_prev) {
                lds_barrier();
                if (uniform(sh->redo[(j - 1) & 3u]) != 0) defer(j - 1);
                check_prev = false;
            }
            defer(j);
            if (rect_ahead) compute_rect(j + 2);
            lds_barrier();  // entry j + 2 is written before anybody reads it (a brick on this branch has no other barrier)
            continue;
        }
        const BrickItem it = item_of(j);
        if (!primed) {
            stage(it, 0, e3, p & 1);
            stores_behind = false;
        }
        primed = false;
        const int64_t n = (static_cast<int64_t>(it.bx * kBrick + lx) * vy + it.by * kBrick + ly) * vz + it.bz * kBrick + lz;
        // voxel of lane (lane & ~3), channel (lane & 3) of a quad: where this lane's 16-byte stores start
        const int64_t quad_off = it.b * C * n_vox + (n - (lane & 3)) + static_cast<int64_t>(lane & 3) * n_vox;
        double X, Y, Z;
        position(it, it.bx * kBrick + lx, it.by * kBrick + ly, it.bz * kBrick + lz, X, Y, Z);

        float2v acc[C / 2];  // sum / conf / max: the aggregate; softmax: the numerator
        float2v den[AGG == MVX_AGG_SOFTMAX ? C / 2 : 1];
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const float z0 = AGG == MVX_AGG_MAX ? -INFINITY : 0.0f;
            acc[i] = float2v{z0, z0};
            if (AGG == MVX_AGG_SOFTMAX) den[i] = float2v{0.0f, 0.0f};
        }
        bool dropped = false;

        for (int v = 0; v < V; ++v) {
            const bool last = v == V - 1;
            // projection into view v, tap addresses in the slot, per-lane check against the staged rectangle
            const int ox = uniform(sh->rect[e3][v].ox), oy = uniform(sh->rect[e3][v].oy), rows = uniform(sh->rect[e3][v].rows);
            const TapXY t = project_tap_xy(sh->cam[e3] + v * 12, X, Y, Z, H, W, wscale);
            const float4 wt = make_float4(t.w00, t.w01, t.w10, t.w11);
            const unsigned any = __builtin_bit_cast(unsigned, t.w00) | __builtin_bit_cast(unsigned, t.w01) |
                                 __builtin_bit_cast(unsigned, t.w10) | __builtin_bit_cast(unsigned, t.w11);
            // voxels whose weights are all zero read the patch origin (always staged) instead of wherever the clamp went
            const int rx = any ? t.xb - ox : 0, ry = any ? t.yb - oy : 0;
            if (!(rx >= 0 && rx + 1 < kPatchW && ry >= 0 && ry + 1 < rows)) sh->bad[e3] = 1;
            const unsigned pixel = lds0 + (p & 1u) * kPatchBytes + static_cast<unsigned>(ry * kPatchW + rx) * kPixelBytes;
            const unsigned a0 = pixel + (static_cast<unsigned>(ry & 7) << 4);
            const unsigned a1 = pixel + kPatchW * kPixelBytes + (static_cast<unsigned>((ry + 1) & 7) << 4);

            // this wave's copies of the unit have landed; then: everybody's have, and the other slot is free
            if (stores_behind)
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            stores_behind = false;
            if (v == 0) {
                // every wave has left brick j - 1: its flags are complete
                if (AGG == MVX_AGG_SOFTMAX && check_prev && uniform(sh->redo[(j - 1) & 3u]) != 0) defer(j - 1);
                check_prev = false;
                if (rect_ahead) compute_rect(j + 2);
            }
            if (last && uniform(sh->bad[e3]) != 0) {
                // some lane's taps lie outside the patches: the brick goes to the direct path, this unit is dropped
                dropped = true;
                ++p;
                break;
            }
            if (!last) {
                stage(it, v + 1, e3, (p + 1) & 1);
            } else if (j + 1 < n_my && uniform(sh->fits[(j + 1) & 3u]) != 0) {
                // (entry j + 1 was written during brick j - 1: at least this unit's barrier lies in between)
                stage(item_of(j + 1), 0, (j + 1) & 3u, (p + 1) & 1);
                primed = true;
            }

            float cconf[C];
            if (AGG == MVX_AGG_CONF) {
#pragma unroll
                for (int c = 0; c < C; ++c) cconf[c] = vconf[(it.b * V + v) * C + c];  // wave-uniform: scalar loads
            }
            float2v chk_d = {0.0f, 0.0f}, chk_r = {0.0f, 0.0f};
            if (MVX_DBG_ON(2u))
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float2v val[2];
                const Taps4 f = patch_taps(a0, a1, q);
                interpolate_x2(f, wt, val[0], val[1]);
                float res[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int cp = q * 2 + h;
                    if (AGG == MVX_AGG_SUM) {
                        acc[cp] = acc[cp] + val[h];
                    } else if (AGG == MVX_AGG_CONF) {
                        acc[cp] = __builtin_elementwise_fma(val[h], float2v{cconf[cp * 2], cconf[cp * 2 + 1]}, acc[cp]);
                    } else if (AGG == MVX_AGG_MAX) {
                        acc[cp] = float2v{fmaxf(acc[cp].x, val[h].x), fmaxf(acc[cp].y, val[h].y)};
                    } else {
                        const float2v e = {__builtin_amdgcn_exp2f(val[h].x), __builtin_amdgcn_exp2f(val[h].y)};
                        den[cp] = den[cp] + e;
                        acc[cp] = __builtin_elementwise_fma(e, val[h], acc[cp]);
                    }
                    if (last) {
                        float2v r = acc[cp];
                        if (AGG == MVX_AGG_SOFTMAX) {
                            const float2v rden = {__builtin_amdgcn_rcpf(den[cp].x), __builtin_amdgcn_rcpf(den[cp].y)};
                            chk_d = chk_d + den[cp];
                            chk_r = chk_r + rden;
                            r = r * rden * float2v{kLn2, kLn2};
                        }
                        res[h * 2] = r.x;
                        res[h * 2 + 1] = r.y;
                    }
                }
                if (last) {
                    // the four lanes of a quad hold four consecutive voxels along z: after the transpose each lane
                    // stores one channel of all four with a single 16-byte store
                    quad_transpose(res, lane);
                    *reinterpret_cast<float4*>(MVX_DBG_PTR(1, vol + quad_off + static_cast<int64_t>(q * 4) * n_vox, 16)) =
                        make_float4(res[0], res[1], res[2], res[3]);
                }
                // keep the quads in order: hoisting the reads of later quads would need hundreds of registers
                __builtin_amdgcn_sched_barrier(0);
            }
            ++p;
            if (last) {
                stores_behind = true;
                if (AGG == MVX_AGG_SOFTMAX) {
                    // every denominator in [2^-80, 2^100] (a NaN fails both comparisons); the flag is read by every wave
                    // after the next barrier
                    if (!(chk_d.x + chk_d.y <= 0x1p100f) || !(chk_r.x + chk_r.y <= 0x1p80f)) sh->redo[e3] = 1;
                    check_prev = true;
                }
            }
        }
        if (dropped) defer(j);
    }

    // ---- the deferred bricks, one at a time through the direct path
    // A brick that failed the softmax check has been stored already and is stored again below by other lanes: the
    // first stores must have left this wave before anybody issues the second ones.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();  // every wave has finished its last unit
    if (AGG == MVX_AGG_SOFTMAX && check_prev && uniform(sh->redo[(n_my - 1) & 3u]) != 0) defer(n_my - 1);
    for (unsigned k = 0; k < n_deferred && MVX_DBG_ON(4u); ++k) {
        lds_barrier();  // the list entry is written, the cameras and tables of the brick before are dead
        const unsigned j = static_cast<unsigned>(uniform(sh->deferred[k]));
        const int64_t b = item_of(j).b;
        for (int e = tid; e < V * 12; e += kBrickThreads) sh->cam[0][e] = camera_entry(proj + (b * V + e / 12) * 12, e % 12, sx, sy);
        lds_barrier();
        direct(j);
    }
}
