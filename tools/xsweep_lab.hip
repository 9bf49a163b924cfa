// xsweep_lab.hip -- prototype (not product code) of the round-4 format for scattered matrices: the X-SWEEP TILE.
//   tile        = a range of rows x one column part; one 16-wavefront workgroup
//   accumulators: one fp32 slot in LDS per pseudo-row (a row of the tile, or one of the k interleaved pieces of a long row);
//                 a slot belongs to ONE wavefront (slot q -> wavefront q mod 16), so it is updated with plain read-add-write
//   x           : the tile's column range is swept in windows of W floats, staged into a ring of three LDS buffers with
//                 coalesced 16-byte loads (every line of x once per tile); step k works on windows k and k+1 while k+2 loads
//   stream      : per wavefront a sequence of 64-element "instructions" {fp32 value, ring index:16 | slot:16}; the packer
//                 schedules the elements so that no two lanes of an instruction update the same slot and an instruction of
//                 step k only touches windows k and k+1
// Measures the kernel on a soc-Pokec-like matrix (bounded power-law row lengths, uniform columns) and checks y.
// Build: hipcc --offload-arch=gfx950 -O3 -fopenmp tools/xsweep_lab.hip -o tools/xsweep_lab
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
#define NWAVES 8
#endif
constexpr int kWaves = NWAVES;
constexpr int kThreads = kWaves * 64;
static int kPiece = 32;          // elements per piece of a long row (set from the number of windows: <= 1/4 element per piece and window)

struct Tile {
    int32_t row0, n_rows;           // rows [row0, row0 + n_rows)
    int32_t col0, n_steps;          // first column of the part (multiple of W), windows to sweep
    int32_t n_slots, slots_per_wave;
    int32_t part, pad;
    int64_t instr_begin[kWaves];    // first instruction of every wavefront's stream (units of 64 elements)
    int64_t cnt_begin;              // counts table: n_steps x 16 bytes
    int64_t qb_begin;               // slot offsets of the rows: n_rows + 1 ints
};

constexpr int kProducers = 4;                      // wavefronts that only stage x windows
constexpr int kAllThreads = (kWaves + kProducers) * 64;

// Consumers (wavefronts 0 .. kWaves-1) run the element stream, producers (the last kProducers wavefronts) stage the x windows:
// a consumer's vector-memory queue then holds nothing but its stream prefetches (a statically unrolled ring, unconditional
// loads: hipcc counts them exactly), a producer's nothing but window loads.  One barrier per step:
//   producer, step k: window k+1 (registers, loaded during step k-1) -> ring[(k+1) % 3]; request window k+2; barrier B_k
//   consumer, step k: barrier B_k; its instructions of step k (they touch windows k and k+1 only)
template <int W>
__global__ __launch_bounds__(kAllThreads) void xsweep_kernel(const Tile* __restrict__ tiles, const uint2* __restrict__ stream,
                                                      const uint8_t* __restrict__ counts, const int32_t* __restrict__ qb,
                                                      const float* __restrict__ x, int cols, float* __restrict__ ypart, int rows) {
    extern __shared__ float lds[];
    float* const win = lds;                 // 3 * W
    float* const acc = lds + 3 * W;         // kWaves * slots_per_wave + 64
    const Tile& T = tiles[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_steps = T.n_steps;
    const int acc_n = T.slots_per_wave * kWaves + 64;
    uint8_t* const cnt = (uint8_t*)(acc + acc_n);          // n_steps * kWaves bytes
    for (int i = threadIdx.x; i < acc_n; i += kAllThreads) acc[i] = 0.0f;
    for (int i = threadIdx.x; i < n_steps * kWaves; i += kAllThreads) cnt[i] = counts[T.cnt_begin + i];
    if (wave >= kWaves) {
        // ---------------- producer
        const int pt = threadIdx.x - kWaves * 64;            // 0 .. kProducers*64-1
        constexpr int kQ = W / (kProducers * 64 * 4);        // float4 per thread and window
#ifndef PREFETCH
#define PREFETCH 3
#endif
        constexpr int PD = PREFETCH;                             // windows in flight (registers) ahead of the one being written
        float4 r[PD][kQ];
        auto load_win = [&](int k, float4* nxt) {
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const long long c = (long long)T.col0 + (long long)k * W + q * (kProducers * 256) + pt * 4;
                nxt[q] = (k < n_steps && c < cols) ? *(const float4*)(x + c) : float4{0, 0, 0, 0};
            }
        };
        auto store_win = [&](int k, const float4* nxt) {
            float* b = win + (k % 3) * W;
#pragma unroll
            for (int q = 0; q < kQ; ++q) *(float4*)(b + q * (kProducers * 256) + pt * 4) = nxt[q];
        };
        load_win(0, r[0]); store_win(0, r[0]);
#pragma unroll
        for (int j = 0; j < PD; ++j) load_win(j + 1, r[j]);      // r[j] holds window j + 1
        for (int k = 0; k < n_steps; k += PD) {
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                if (k + j < n_steps) {
                    store_win(k + j + 1, r[j]);                  // requested PD steps ago
                    load_win(k + j + 1 + PD, r[j]);
                    __syncthreads();
                }
            }
        }
    } else {
        // ---------------- consumer
        const uint2* sp = stream + (size_t)T.instr_begin[wave] * 64 + lane;
        int total = 0;
        __syncthreads();                                      // B_0: counts, accumulators, windows 0 and 1 are in the LDS
        for (int k = 0; k < n_steps; ++k) total += cnt[k * kWaves + wave];
        total = __builtin_amdgcn_readfirstlane(total);
        // The stream ring: D instructions in flight per wavefront, loads issued with inline asm and awaited with an explicit
        // s_waitcnt vmcnt(D-1): hipcc's own wait insertion drains the queue once per trip of a loop like this one (vmcnt(1)
        // before the oldest buffer), which exposes an HBM latency every D instructions -- and with one barrier per step the
        // whole workgroup then runs at the pace of whichever wavefront is draining.
        constexpr int D = 8;
        unsigned long long buf[D];
        auto issue = [&](unsigned long long& dst, const uint2* ptr) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory"); };
#pragma unroll
        for (int u = 0; u < D; ++u) issue(buf[u], sp + (size_t)u * 64);           // (the stream has D instructions of slack behind its end)
        int k = 0, remaining = __builtin_amdgcn_readfirstlane((int)cnt[wave]);
        for (int pos = 0; pos < total; pos += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                if (remaining == 0) { ++k; __syncthreads(); remaining = __builtin_amdgcn_readfirstlane((int)cnt[k * kWaves + wave]); }
                asm volatile("s_waitcnt vmcnt(7)" : "+v"(buf[u]) :: "memory");
                const unsigned ev = (unsigned)buf[u], em = (unsigned)(buf[u] >> 32);
                const float xv = win[em & 0xffffu];
                const unsigned slot = em >> 16;
                const float p = __builtin_bit_cast(float, ev) * xv;
                const float a = acc[slot];
                acc[slot] = a + p;
                --remaining;
                issue(buf[u], sp + (size_t)(pos + u + D) * 64);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7]) :: "memory");
        while (k < n_steps - 1) { ++k; __syncthreads(); }   // every wavefront passes the same number of barriers
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
    int rows = 1632803, W = 8192, parts = 2, tile_slots = 12288;
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

    // --- tiles: column parts of equal width (multiple of W); row ranges with <= tile_slots slots and ~ equal elements
    const int part_w = (int)((((int64_t)cols + parts - 1) / parts + W - 1) / W * W);
    kPiece = std::max(4, std::min(32, part_w / W / 4));
    printf("piece = %d elements\n", kPiece);
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
    struct Packed { std::vector<uint2> instr[kWaves]; std::vector<uint8_t> counts; std::vector<int32_t> qb; int64_t pad = 0, elems = 0; };
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
        P.counts.assign((size_t)T.n_steps * kWaves, 0);
        for (int w = 0; w < kWaves; ++w) {
            auto& E = we[w];
            std::stable_sort(E.begin(), E.end(), [](const El& a, const El& b2) { return a.win < b2.win; });
            const int S = T.slots_per_wave;
            // step k may use windows k (last chance) and k + 1.  The mandatory elements open as many instructions as they need
            // (no two lanes of an instruction on the same slot), the optional ones fill them; further instructions only full ones.
            std::vector<El> M, O, rest;
            size_t next = 0;
            auto emit = [&](std::vector<El>& ins, int k) {
                for (size_t l = 0; l < 64; ++l) {
                    if (l < ins.size()) {
                        const El& e = ins[l];
                        const uint32_t ring = (uint32_t)((e.win % 3) * W + (e.col - c0) % W);
                        const uint32_t slot = (uint32_t)(w * S + e.q / kWaves);
                        uint32_t vb; memcpy(&vb, &e.v, 4);
                        P.instr[w].push_back(uint2{vb, ring | (slot << 16)});
                    } else { P.instr[w].push_back(uint2{0u, (uint32_t)(kWaves * S + l) << 16}); P.pad++; }
                }
                if (++P.counts[(size_t)k * kWaves + w] == 255) { fprintf(stderr, "count overflow\n"); exit(1); }
            };
            for (int k = 0; k < T.n_steps; ++k) {
                M.swap(O); O.clear();                  // what was optional is mandatory now
                if (k == 0) { while (next < E.size() && E[next].win <= 0) M.push_back(E[next++]); }
                while (next < E.size() && E[next].win <= k + 1) O.push_back(E[next++]);
                std::vector<std::vector<El>> ins;
                auto place = [&](const El& e) {
                    for (auto& I : ins) {
                        if (I.size() >= 64) continue;
                        bool clash = false;
                        for (const El& o : I) if (o.q == e.q) { clash = true; break; }
                        if (!clash) { I.push_back(e); return true; }
                    }
                    return false;
                };
                for (const El& e : M) if (!place(e)) { ins.emplace_back(); ins.back().push_back(e); }
                rest.clear();
                for (const El& e : O) if (!place(e)) rest.push_back(e);
                while (rest.size() >= 64) {               // more full instructions from the optional elements
                    std::vector<El> I, left;
                    for (const El& e : rest) {
                        bool clash = I.size() >= 64;
                        for (size_t z = 0; !clash && z < I.size(); ++z) clash = I[z].q == e.q;
                        if (clash) left.push_back(e); else I.push_back(e);
                    }
                    if (I.size() < 64) break;
                    ins.push_back(std::move(I)); rest.swap(left);
                }
                O = rest;
                if (ins.empty()) ins.emplace_back();     // every wavefront has >= 1 instruction in every step: the consumer passes ONE barrier per step change
                for (auto& I : ins) emit(I, k);
                P.elems += 0;
            }
            while ((P.instr[w].size() / 64) % 8 != 0) {      // whole groups of D = 8 instructions: the consumer loop has no remainder branch
                std::vector<El> none; emit(none, T.n_steps - 1);
            }
            P.elems += (int64_t)E.size();
            if (false && t == 5 && w < 3) { fprintf(stderr, "tile 5 wave %d: %zu elements, %zu instr; steps:", w, E.size(), P.instr[w].size() / 64); for (int k = 0; k < T.n_steps; ++k) { int c = 0; for (auto& e : E) c += e.win == k; fprintf(stderr, " %d/%d", c, (int)P.counts[(size_t)k * kWaves + w]); } fprintf(stderr, "\n"); }
            if (!O.empty()) { fprintf(stderr, "elements left over\n"); exit(1); }
        }
    }
    printf("packed in %.1f s\n", omp_get_wtime() - t0);
    // --- concatenate
    std::vector<uint2> stream; std::vector<uint8_t> counts; std::vector<int32_t> qb;
    int64_t pad = 0, elems = 0; int max_spw = 0, max_steps = 0;
    for (int t = 0; t < n_tiles; ++t) {
        Tile& T = tiles[t]; Packed& P = packed[t];
        for (int w = 0; w < kWaves; ++w) { T.instr_begin[w] = (int64_t)stream.size() / 64; stream.insert(stream.end(), P.instr[w].begin(), P.instr[w].end()); }
        T.cnt_begin = (int64_t)counts.size(); counts.insert(counts.end(), P.counts.begin(), P.counts.end());
        T.qb_begin = (int64_t)qb.size(); qb.insert(qb.end(), P.qb.begin(), P.qb.end());
        pad += P.pad; elems += P.elems; max_spw = std::max(max_spw, T.slots_per_wave); max_steps = std::max(max_steps, T.n_steps);
    }
    printf("stream %.1f MB (%lld elements + %lld padding = %.2f %%), counts %.2f MB, qb %.1f MB, max slots per wave %d, steps %d\n", stream.size() * 8 / 1e6,
           (long long)elems, (long long)pad, 100.0 * pad / std::max<int64_t>(1, elems), counts.size() / 1e6, qb.size() * 4 / 1e6, max_spw, max_steps);
    const size_t lds = ((size_t)3 * W + (size_t)max_spw * kWaves + 64) * 4 + (size_t)max_steps * kWaves + 64;
    printf("LDS per workgroup %.1f KiB\n", lds / 1024.0);
    if (lds > 160 * 1024 - 256) { printf("does not fit\n"); return 1; }
    // longest tiles first
    std::vector<int> order((size_t)n_tiles);
    for (int i = 0; i < n_tiles; ++i) order[i] = i;
    std::vector<int64_t> work((size_t)n_tiles);
    for (int t = 0; t < n_tiles; ++t) { work[t] = 0; for (int w = 0; w < kWaves; ++w) work[t] = std::max<int64_t>(work[t], (int64_t)packed[t].instr[w].size()); }
    // (256 tiles on 256 CUs: all resident at once; the order IS the XCD mapping)
    std::vector<Tile> sorted;
    for (int i : order) sorted.push_back(tiles[i]);

    Tile* d_tiles; uint2* d_stream; uint8_t* d_counts; int32_t* d_qb; float *d_x, *d_y;
    CK(hipMalloc(&d_tiles, sorted.size() * sizeof(Tile))); CK(hipMemcpy(d_tiles, sorted.data(), sorted.size() * sizeof(Tile), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_stream, stream.size() * 8 + 64 * 512)); CK(hipMemcpy(d_stream, stream.data(), stream.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_counts, counts.size() + 64)); CK(hipMemcpy(d_counts, counts.data(), counts.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_qb, qb.size() * 4)); CK(hipMemcpy(d_qb, qb.data(), qb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_y, (size_t)parts * rows * 4)); CK(hipMemset(d_y, 0xff, (size_t)parts * rows * 4));
    auto launch = [&]() {
        if (W == 4096) hipLaunchKernelGGL(xsweep_kernel<4096>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_counts, d_qb, d_x, cols, d_y, rows);
        else if (W == 8192) hipLaunchKernelGGL(xsweep_kernel<8192>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_counts, d_qb, d_x, cols, d_y, rows);
        else if (W == 3072) hipLaunchKernelGGL(xsweep_kernel<3072>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_counts, d_qb, d_x, cols, d_y, rows);
        else if (W == 6144) hipLaunchKernelGGL(xsweep_kernel<6144>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_counts, d_qb, d_x, cols, d_y, rows);
        else if (W == 12288) hipLaunchKernelGGL(xsweep_kernel<12288>, dim3(n_tiles), dim3(kAllThreads), lds, 0, d_tiles, d_stream, d_counts, d_qb, d_x, cols, d_y, rows);
        else { printf("unsupported W\n"); exit(1); }
    };
    CK(hipFuncSetAttribute((const void*)xsweep_kernel<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)xsweep_kernel<8192>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)xsweep_kernel<3072>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)xsweep_kernel<6144>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    CK(hipFuncSetAttribute((const void*)xsweep_kernel<12288>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
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
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("kernel %.1f us  (%.1f Gelem/s, algorithmic %.0f GB/s)\n", ms * 1e3, nnz / ms / 1e6, (8.0 * nnz + 16.0 * rows) / ms / 1e6);
    }
    return bad != 0;
}
