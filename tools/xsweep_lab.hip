// xsweep_lab.hip -- prototype (not product code) of the round-4 format for scattered matrices: the X-SWEEP TILE.
//   tile        = a range of rows x one column part; one workgroup of NWAVES consumer + 4 producer wavefronts, one per CU
//   accumulators: one fp32 slot in LDS per pseudo-row (a row of the tile, or one of the k interleaved pieces of a long row);
//                 a slot belongs to ONE consumer wavefront (slot q -> wavefront q mod NWAVES): plain read-add-write, no atomics
//   x           : the tile's column range is swept in windows of W floats through a ring of RING LDS buffers, staged by the
//                 producer wavefronts with coalesced 16-byte loads (every line of x once per tile)
//   stream      : per consumer wavefront a sequence of 64-element instructions {fp32 value, ring index:16 | slot:16} + a header
//                 {lo, hi} = the windows it touches (hi - lo <= SPAN); the packer schedules the elements so that no two lanes
//                 of an instruction update the same slot
//   sync        : no barrier inside the sweep -- ready[p] (windows producer p has written), done[w] (windows consumer w no
//                 longer needs) in LDS; a consumer waits for ready > hi, a producer for done >= j - RING + 1 before it overwrites
// Measures the kernel on a soc-Pokec-like matrix (bounded power-law row lengths, uniform columns) and checks y.
// Build: hipcc --offload-arch=gfx950 -O3 -fopenmp -ffp-contract=off -DNWAVES=12 -DRING=8 tools/xsweep_lab.hip -o tools/xsweep_lab
#include <hip/hip_runtime.h>
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#ifndef NWAVES
#define NWAVES 12
#endif
#ifndef RING
#define RING 8
#endif
#ifndef PREFETCH
#define PREFETCH 4
#endif
constexpr int kWaves = NWAVES;                      // consumer wavefronts
constexpr int kProducers = 4;                       // wavefronts that only stage x windows
constexpr int kAllThreads = (kWaves + kProducers) * 64;
constexpr int kSpan = RING - 2;                     // an instruction touches windows lo .. hi, hi - lo <= kSpan
static int kPiece = 32;                             // elements per piece of a long row

struct Tile {
    int32_t row0, n_rows;           // rows [row0, row0 + n_rows)
    int32_t col0, n_steps;          // first column of the part (multiple of W), windows to sweep
    int32_t n_slots, slots_per_wave;
    int32_t part, n_instr;          // instructions of the tile (all wavefronts)
    int64_t instr_begin[kWaves + 1];    // first instruction of every wavefront's stream (units of 64 elements); [kWaves] = end
    int64_t qb_begin;               // slot offsets of the rows: n_rows + 1 ints
};

template <int W>
__global__ __launch_bounds__(kAllThreads) void xsweep_kernel(const Tile* __restrict__ tiles, const uint2* __restrict__ stream,
                                                      const uint32_t* __restrict__ hdrs, const int32_t* __restrict__ qb,
                                                      const float* __restrict__ x, int cols, float* __restrict__ ypart, int rows, int mode) {
    // mode 0: the kernel; 1: producers only (consumers declare every window free at once); 2: consumers only (producers declare every
    // window written without loading it): the floors of the two sides
    extern __shared__ float lds[];
    float* const win = lds;                 // RING * W
    float* const acc = lds + RING * W;      // kWaves * slots_per_wave + 64
    const Tile& T = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_steps = T.n_steps;
    const int acc_n = T.slots_per_wave * kWaves + 64;
    volatile int* const flags = (volatile int*)(acc + acc_n);        // ready[4], done[16]
    uint32_t* const hdr = (uint32_t*)(acc + acc_n + 32);             // one per instruction of the tile: lo | hi << 16
    for (int i = threadIdx.x; i < acc_n; i += kAllThreads) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < T.n_instr; i += kAllThreads) hdr[i] = hdrs[T.instr_begin[0] + i];
    if (threadIdx.x < 32) flags[threadIdx.x] = threadIdx.x >= 4 + kWaves && threadIdx.x < 20 ? 0x7fffffff : 0;      // (unused done[] entries never hold a producer back)
    __syncthreads();
    if (wave >= kWaves) {
        // ---------------- producer
        const int pw = wave - kWaves;
        if (mode == 2) { if (lane == 0) flags[pw] = 0x7ffffff0; }
        else {
        const int pt = threadIdx.x - kWaves * 64;            // 0 .. kProducers*64-1
        constexpr int kQ = W / (kProducers * 64 * 4);        // float4 per thread and window
        constexpr int PD = PREFETCH;                         // windows in flight (registers)
        float4 r[PD][kQ];
        auto load_win = [&](int k, float4* nxt) {
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const long long c = (long long)T.col0 + (long long)k * W + q * (kProducers * 256) + pt * 4;
                nxt[q] = (k < n_steps && c < cols) ? *(const float4*)(x + c) : float4{0, 0, 0, 0};
            }
        };
        auto store_win = [&](int k, const float4* nxt) {
            float* b = win + (k % RING) * W;
#pragma unroll
            for (int q = 0; q < kQ; ++q) *(float4*)(b + q * (kProducers * 256) + pt * 4) = nxt[q];
        };
#pragma unroll
        for (int j = 0; j < PD; ++j) load_win(j, r[j]);          // r[j] holds window j
        int freed = 0;                                           // windows below this are no longer needed by any consumer
        for (int j0 = 0; j0 < n_steps; j0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int j = j0 + u;
                if (j < n_steps) {
                    while (freed < j - RING + 1) {
                        int m = 0x7fffffff;
#pragma unroll
                        for (int q = 0; q < 16; ++q) m = min(m, flags[4 + q]);
                        freed = __builtin_amdgcn_readfirstlane(m);
                        if (freed < j - RING + 1) __builtin_amdgcn_s_sleep(2);
                    }
                    store_win(j, r[u]);
                    load_win(j + PD, r[u]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) flags[pw] = j + 1;
                }
            }
        }
        }
    } else {
        // ---------------- consumer
        if (mode == 1) { if (lane == 0) flags[4 + wave] = 0x7fffffff; }
        else {
        const int i0 = (int)(T.instr_begin[wave] - T.instr_begin[0]);
        const int total = (int)(T.instr_begin[wave + 1] - T.instr_begin[wave]);          // a multiple of D
        const uint2* sp = stream + (size_t)T.instr_begin[wave] * 64 + lane;
        constexpr int D = 8;
        unsigned long long buf[D];
        auto issue = [&](unsigned long long& dst, const uint2* ptr) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); };
#pragma unroll
        for (int u = 0; u < D; ++u) issue(buf[u], sp + (size_t)u * 64);           // (the stream has D instructions of slack behind its end)
        int seen = 0, published = 0;
        for (int pos = 0; pos < total; pos += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const unsigned h = __builtin_amdgcn_readfirstlane(hdr[i0 + pos + u]);
                const int lo = (int)(h & 0xffffu), hi = (int)(h >> 16);
                if (lo > published) {                            // windows below lo are free
                    published = lo;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) flags[4 + wave] = lo;
                }
                while (seen <= hi) {                             // window hi must have been written by all producers
                    seen = __builtin_amdgcn_readfirstlane(min(min(flags[0], flags[1]), min(flags[2], flags[3])));
                    if (seen <= hi) __builtin_amdgcn_s_sleep(1);
                    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(buf[u]) :: "memory");
                const unsigned ev = (unsigned)buf[u], em = (unsigned)(buf[u] >> 32);
                const float xv = win[em & 0xffffu];
                const unsigned slot = em >> 16;
                const float p = __builtin_bit_cast(float, ev) * xv;
                const float a = acc[slot];
                acc[slot] = a + p;
                issue(buf[u], sp + (size_t)(pos + u + D) * 64);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]) :: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) flags[4 + wave] = 0x7fffffff;                 // done with every window
        }
    }
    __syncthreads();
    // rows of the tile: the pieces of a row are consecutive slots q; slot q lives at (q % kWaves) * S + q / kWaves
    const int S = T.slots_per_wave;
    const int32_t* tq = qb + T.qb_begin;
    float* yo = ypart + (size_t)T.part * rows + T.row0;
    for (int i = threadIdx.x; i < T.n_rows; i += kAllThreads) {
        const int q0 = tq[i], q1 = tq[i + 1];
        float s = acc[(q0 % kWaves) * S + q0 / kWaves];
        for (int q = q0 + 1; q < q1; ++q) s += acc[(q % kWaves) * S + q / kWaves];
        yo[i] = s;
    }
}

int main(int argc, char** argv) {
    int rows = 1632803, W = 3072, parts = 2, tile_slots = 0;
    double avg = 18.75;
    if (argc > 1) W = atoi(argv[1]);
    if (argc > 2) parts = atoi(argv[2]);
    if (argc > 3) tile_slots = atoi(argv[3]);
    if (argc > 4) rows = atoi(argv[4]);
    const int cols = rows;
    const int64_t nnz_target = (int64_t)(rows * avg);
    printf("rows %d, window %d floats, %d column parts, <= %d slots per tile\n", rows, W, parts, tile_slots);
    // --- matrix: bounded power-law row lengths (as hispmv_amd/matrices.py), uniform columns, sorted per row
    std::mt19937_64 rng(7);
    std::vector<int64_t> rp((size_t)rows + 1, 0);
    {
        std::vector<double> w((size_t)rows);
        std::vector<int> perm((size_t)rows);
        for (int i = 0; i < rows; ++i) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), rng);
        double sum = 0;
        for (int i = 0; i < rows; ++i) { w[i] = 1.0 / std::pow(perm[i] + 100.0, 0.8); sum += w[i]; }
        for (int i = 0; i < rows; ++i) { std::poisson_distribution<int> pd(w[i] * nnz_target / sum); rp[(size_t)i + 1] = rp[i] + pd(rng); }
    }
    const int64_t nnz = rp[rows];
    std::vector<int32_t> ci((size_t)nnz); std::vector<float> va((size_t)nnz);
    {
        int maxlen = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(max : maxlen)
        for (int i = 0; i < rows; ++i) {
            std::mt19937_64 g(1000 + i);
            for (int64_t k = rp[i]; k < rp[(size_t)i + 1]; ++k) { ci[k] = (int32_t)(g() % (uint64_t)cols); va[k] = (float)((g() % 2000) / 1000.0 - 1.0); }
            std::sort(ci.begin() + rp[i], ci.begin() + rp[(size_t)i + 1]);
            maxlen = std::max(maxlen, (int)(rp[(size_t)i + 1] - rp[i]));
        }
        printf("nnz %lld, longest row %d\n", (long long)nnz, maxlen);
    }
    std::vector<float> x((size_t)cols + 4);
    for (int j = 0; j < cols; ++j) x[j] = (float)((j % 1000) / 1000.0 + 0.001);

    // --- tiles: column parts of equal width (multiple of W); exactly n_cus tiles
    const int part_w = (int)((((int64_t)cols + parts - 1) / parts + W - 1) / W * W);
    kPiece = std::max(4, std::min(32, part_w / W / 4));
    printf("piece = %d elements, span = %d windows\n", kPiece, kSpan);
    std::vector<Tile> tiles;
    const int n_cus = 256;
    const int row_tiles_wanted = std::max(1, n_cus / parts);
    {   // exactly row_tiles_wanted row ranges of ~equal elements; tile b runs on XCD b % 8 (workgroups are dealt round-robin over the
        // XCDs): the tiles of a column part go to the same 8 / parts XCDs, so that the 32 CUs of an XCD sweep the SAME part of x at
        // about the same pace and its L2 serves the windows (x is larger than one L2)
        std::vector<int> cut((size_t)row_tiles_wanted + 1, 0);
        int r = 0;
        for (int i = 1; i <= row_tiles_wanted; ++i) {
            const int64_t target = nnz * i / row_tiles_wanted;
            while (r < rows && rp[(size_t)r + 1] <= target) ++r;
            if (i == row_tiles_wanted) r = rows;
            cut[i] = std::max(r, cut[i - 1]);
        }
        const int per = 8 / parts > 0 ? 8 / parts : 1;
        for (int b = 0; b < row_tiles_wanted * parts; ++b) {
            const int xcd = b % 8, p = parts >= 8 ? xcd : xcd / per;
            const int range = parts >= 8 ? b / 8 : (b / 8) * per + xcd % per;
            Tile t{}; t.row0 = cut[range]; t.n_rows = cut[range + 1] - cut[range]; t.col0 = p * part_w; t.part = p;
            tiles.push_back(t);
        }
        (void)tile_slots;
    }
    const int n_tiles = (int)tiles.size();
    printf("%d tiles (%d row ranges)\n", n_tiles, n_tiles / parts);
    // --- pack every tile
    struct Packed { std::vector<uint2> instr[kWaves]; std::vector<uint32_t> hdr[kWaves]; std::vector<int32_t> qb; int64_t pad = 0, elems = 0; };
    std::vector<Packed> packed((size_t)n_tiles);
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 1)
    for (int t = 0; t < n_tiles; ++t) {
        Tile& T = tiles[t];
        Packed& P = packed[t];
        const int c0 = T.col0, c1 = std::min(cols, T.col0 + part_w);
        T.n_steps = (c1 - c0 + W - 1) / W;
        struct El { int32_t win; int32_t q; int32_t col; float v; };
        std::vector<El> we[kWaves];
        P.qb.assign((size_t)T.n_rows + 1, 0);
        int q = 0;
        for (int i = 0; i < T.n_rows; ++i) {
            const int r = T.row0 + i;
            const int32_t* b = ci.data() + rp[r]; const int32_t* e = ci.data() + rp[(size_t)r + 1];
            const int64_t k0 = std::lower_bound(b, e, c0) - ci.data(), k1 = std::lower_bound(b, e, c1) - ci.data();
            const int n = (int)(k1 - k0);
            const int m = std::max(1, (n + kPiece - 1) / kPiece);
            P.qb[i] = q;
            for (int j = 0; j < n; ++j) {
                const int qq = q + j % m;
                we[qq % kWaves].push_back(El{(ci[k0 + j] - c0) / W, qq, ci[k0 + j], va[k0 + j]});
            }
            q += m;
        }
        P.qb[T.n_rows] = q;
        T.n_slots = q; T.slots_per_wave = (q + kWaves - 1) / kWaves;
        const int S = T.slots_per_wave;
        for (int w = 0; w < kWaves; ++w) {
            auto& E = we[w];
            std::stable_sort(E.begin(), E.end(), [](const El& a, const El& b2) { return a.win < b2.win; });
            P.elems += (int64_t)E.size();
            // greedy: an instruction takes the oldest unscheduled elements, at most 64, windows within first .. first + kSpan,
            // no slot twice (an element whose slot is taken stays for the next instruction)
            std::vector<char> taken(E.size(), 0);
            std::vector<int32_t> mark((size_t)S, -1);
            size_t head = 0; int ino = 0;
            auto emit = [&](const std::vector<size_t>& pick, int lo, int hi) {
                for (size_t l = 0; l < 64; ++l) {
                    if (l < pick.size()) {
                        const El& e = E[pick[l]];
                        const uint32_t ring = (uint32_t)((e.win % RING) * W + (e.col - c0) % W);
                        const uint32_t slot = (uint32_t)(w * S + e.q / kWaves);
                        uint32_t vb; memcpy(&vb, &e.v, 4);
                        P.instr[w].push_back(uint2{vb, ring | (slot << 16)});
                    } else { P.instr[w].push_back(uint2{0u, (uint32_t)(kWaves * S + l) << 16}); P.pad++; }
                }
                P.hdr[w].push_back((uint32_t)lo | ((uint32_t)hi << 16));
            };
            int last_lo = 0;
            while (head < E.size()) {
                while (head < E.size() && taken[head]) ++head;
                if (head >= E.size()) break;
                const int lo = E[head].win;
                std::vector<size_t> pick;
                int hi = lo;
                for (size_t z = head; z < E.size() && pick.size() < 64 && E[z].win <= lo + kSpan; ++z) {
                    if (taken[z]) continue;
                    const int ls = E[z].q / kWaves;
                    if (mark[ls] == ino) continue;
                    mark[ls] = ino; taken[z] = 1; pick.push_back(z); hi = std::max(hi, (int)E[z].win);
                }
                emit(pick, lo, hi); ++ino; last_lo = lo;
            }
            while ((P.instr[w].size() / 64) % 8 != 0) { std::vector<size_t> none; emit(none, last_lo, last_lo); }     // whole groups of D = 8 instructions
        }
    }
    printf("packed in %.1f s\n", omp_get_wtime() - t0);
    // --- concatenate
    std::vector<uint2> stream; std::vector<uint32_t> hdrs; std::vector<int32_t> qb;
    int64_t pad = 0, elems = 0; int max_spw = 0, max_steps = 0, max_instr = 0;
    for (int t = 0; t < n_tiles; ++t) {
        Tile& T = tiles[t]; Packed& P = packed[t];
        for (int w = 0; w < kWaves; ++w) {
            T.instr_begin[w] = (int64_t)stream.size() / 64;
            stream.insert(stream.end(), P.instr[w].begin(), P.instr[w].end());
            hdrs.insert(hdrs.end(), P.hdr[w].begin(), P.hdr[w].end());
        }
        T.instr_begin[kWaves] = (int64_t)stream.size() / 64;
        T.n_instr = (int)(T.instr_begin[kWaves] - T.instr_begin[0]);
        T.qb_begin = (int64_t)qb.size(); qb.insert(qb.end(), P.qb.begin(), P.qb.end());
        pad += P.pad; elems += P.elems; max_spw = std::max(max_spw, T.slots_per_wave); max_steps = std::max(max_steps, T.n_steps); max_instr = std::max(max_instr, T.n_instr);
    }
    printf("stream %.1f MB (%lld elements + %lld padding = %.2f %%), headers %.2f MB, qb %.1f MB, max slots per wave %d, steps %d, max instructions per tile %d\n", stream.size() * 8 / 1e6,
           (long long)elems, (long long)pad, 100.0 * pad / std::max<int64_t>(1, elems), hdrs.size() * 4 / 1e6, qb.size() * 4 / 1e6, max_spw, max_steps, max_instr);
    const size_t lds = ((size_t)RING * W + (size_t)max_spw * kWaves + 64 + 32 + (size_t)max_instr) * 4 + 64;
    printf("LDS per workgroup %.1f KiB\n", lds / 1024.0);
    if (lds > 160 * 1024 - 256) { printf("does not fit\n"); return 1; }

    Tile* d_tiles; uint2* d_stream; uint32_t* d_hdrs; int32_t* d_qb; float *d_x, *d_y;
    CK(hipMalloc(&d_tiles, tiles.size() * sizeof(Tile))); CK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(Tile), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_stream, stream.size() * 8 + 64 * 512)); CK(hipMemcpy(d_stream, stream.data(), stream.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_hdrs, hdrs.size() * 4 + 64)); CK(hipMemcpy(d_hdrs, hdrs.data(), hdrs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_qb, qb.size() * 4)); CK(hipMemcpy(d_qb, qb.data(), qb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)parts * rows * 4)); CK(hipMemset(d_y, 0xff, (size_t)parts * rows * 4));
#define LAUNCH(WW) hipLaunchKernelGGL(xsweep_kernel<WW>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_hdrs, d_qb, d_x, cols, d_y, rows, mode)
    int mode = 0;
    auto launch = [&]() {
        if (W == 1024) LAUNCH(1024); else if (W == 2048) LAUNCH(2048); else if (W == 3072) LAUNCH(3072); else if (W == 4096) LAUNCH(4096);
        else if (W == 6144) LAUNCH(6144); else if (W == 8192) LAUNCH(8192); else { printf("unsupported W\n"); exit(1); }
    };
#define RAISE(WW) CK(hipFuncSetAttribute((const void*)xsweep_kernel<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256))
    RAISE(1024); RAISE(2048); RAISE(3072); RAISE(4096); RAISE(6144); RAISE(8192);
    launch(); CK(hipDeviceSynchronize());
    // --- check
    std::vector<float> y((size_t)parts * rows);
    CK(hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; int bad = 0;
    for (int i = 0; i < rows; ++i) {
        double ref = 0, mag = 0;
        for (int64_t k = rp[i]; k < rp[(size_t)i + 1]; ++k) { ref += (double)va[k] * x[ci[k]]; mag += std::fabs((double)va[k] * x[ci[k]]); }
        double got = 0;
        for (int p = 0; p < parts; ++p) got += y[(size_t)p * rows + i];
        const double err = std::fabs(got - ref) / std::max(mag, 1e-30);
        if (!(err < 1e-5) && mag > 0) { if (bad < 5) printf("row %d: got %g ref %g\n", i, got, ref); ++bad; }
        if (mag > 0) worst = std::max(worst, err);
    }
    printf("check: worst backward error %.3g, %d bad rows\n", worst, bad);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int md : {0, 0, 1, 2}) {
        mode = md;
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%s %.1f us  (%.1f Gelem/s, algorithmic %.0f GB/s)\n", md == 0 ? "kernel" : md == 1 ? "producers only" : "consumers only", ms * 1e3, nnz / ms / 1e6, (8.0 * nnz + 16.0 * rows) / ms / 1e6);
    }
    return bad != 0;
}
