// hispmv_prep_abi.cpp -- the host-only hispmv_prep_* entry points of include/hispmv.h: the preprocessor, the launch planner, the
// format / tiling choice and the tile-stream packer without a device, for tests on a CPU-only box.  Mirrors
// HiSpmvHandle::getPreparedMtx (common/src/spmv-helper.cpp:800-802).  (hispmv_prep_from_coo_device is the one entry here that
// touches the GPU.)
#include "hispmv_ctx.h"

using namespace hispmv;

#define g_prep_err (::hispmv::prep_error())

// ---- host-only preprocessor access ---------------------------------------------------------------
HISPMV_API const char* hispmv_prep_last_error(void) { return g_prep_err.c_str(); }

HISPMV_API int hispmv_host_threads(void) { return host_threads(); }

HISPMV_API int hispmv_prep_from_coo(hispmv_prep** out, const int32_t* r, const int32_t* cl, const float* v,
                                    int64_t nnz, int32_t rows, int32_t cols) {
    if (!out) return HISPMV_EINVAL;
    host_threads();
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || nnz < 0) { g_prep_err = "bad sparse matrix arguments"; return HISPMV_EINVAL; }
    try {
        auto p = std::make_unique<hispmv_prep>();
        p->csr = coo_to_csr(rows, cols, nnz, r, cl, v);
        p->st = build_stream(p->csr);
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_from_coo_device(hispmv_prep** out, int device_id, const int32_t* r, const int32_t* cl, const float* v,
                                           int64_t nnz, int32_t rows, int32_t cols, double seconds[5]) {
    if (!out) return HISPMV_EINVAL;
    *out = nullptr;
    if (rows <= 0 || cols <= 0 || nnz < 0 || (nnz > 0 && (!r || !cl || !v))) { g_prep_err = "bad sparse matrix arguments"; return HISPMV_EINVAL; }
    if (hipSetDevice(device_id) != hipSuccess) { g_prep_err = "no such HIP device"; return HISPMV_EDEVICE; }
    try {
        auto p = std::make_unique<hispmv_prep>();
        DevicePrepTimes t;
        std::string err;
        if (!prep_on_device(rows, cols, nnz, r, cl, v, p->csr, p->st, t, err)) {
            g_prep_err = err;
            return err.find("outside") != std::string::npos || err.find("dimension") != std::string::npos ? HISPMV_EINVAL : HISPMV_EDEVICE;
        }
        if (seconds) { seconds[0] = t.upload; seconds[1] = t.csr_device; seconds[2] = t.offsets_host; seconds[3] = t.stream_device; seconds[4] = t.download; }
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_from_mtx(hispmv_prep** out, const char* path, int flavor) {
    host_threads();
    if (!out) return HISPMV_EINVAL;
    *out = nullptr;
    if (!path || (flavor != 0 && flavor != 1)) { g_prep_err = "bad arguments"; return HISPMV_EINVAL; }
    try {
        Coo coo = read_mtx(path, (MtxFlavor)flavor);
        auto p = std::make_unique<hispmv_prep>();
        p->csr = coo_to_csr(coo.rows, coo.cols, (int64_t)coo.r.size(), coo.r.data(), coo.c.data(), coo.v.data());
        p->st = build_stream(p->csr);
        *out = p.release();
        return HISPMV_OK;
    } catch (const std::runtime_error& ex) { g_prep_err = ex.what(); return HISPMV_EIO;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API void hispmv_prep_free(hispmv_prep* p) { delete p; }

// The format / tiling decision of the loader for this matrix on a device with n_cus compute units, host-only
// (hispmv_choose.cpp: the same function hispmv_create_sparse_handle* calls).  Works on a copy of the prepared CSR.
HISPMV_API int hispmv_prep_choose_format(const hispmv_prep* p, int n_cus, int64_t out[16]) {
    if (!p || !out || n_cus <= 0) return HISPMV_EINVAL;
    try {
        Csr copy = p->csr;
        FormatOptions opt = FormatOptions::from_env();
        opt.decide_only = true;
        const FormatChoice ch = choose_format(std::move(copy), nullptr, n_cus, opt);
        int64_t n_slices = 0, n_elems = 0, n_split = 0, global_elems = 0;
        for (const HostPart& q : ch.parts) {
            if (q.is_tts) { n_slices += (int64_t)q.tts.col_base.size(); n_elems += q.tts.nnz + q.tts.n_fillers; n_split += (int64_t)q.tts.fix.size() / 4; }
            else { n_slices += q.st.n_slices; n_elems += q.st.n_elems; n_split += (int64_t)q.st.fix.size(); global_elems += q.plan.lds_floats > 0 ? q.plan.global_elems : q.st.n_slices * (int64_t)kSliceElems; }
        }
        const HostPart& p0 = ch.parts[0];
        out[0] = ch.format; out[1] = ch.parts.size() > 1 ? (ch.tile_kind ? ch.tile_kind : 1) : 0; out[2] = (int64_t)ch.parts.size();
        out[3] = ch.col_tile_width; out[4] = ch.col_tile_base; out[5] = ch.l2_tiles ? 1 : 0;
        out[6] = ch.format == 1 ? p0.tts.geometry.threads : p0.plan.block_threads;
        out[7] = ch.format == 1 ? p0.tts.geometry.max_slots / kTtsChunk : p0.plan.group_slices;
        out[8] = ch.format == 1 ? 0 : p0.plan.lds_floats;
        out[9] = n_slices; out[10] = n_elems; out[11] = n_split;
        out[12] = (int64_t)(ch.tts_lines_per_gather * 1000.0 + 0.5);
        out[13] = global_elems;        // elements that gather x through L2 (slice streams: outside their window, or no window at all)
        out[14] = 0; out[15] = 0;
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_step_queue(const double* slice_costs, int32_t n_slice, const double* tile_costs, int32_t n_tile, int32_t n_wg, int32_t mode,
                                      int32_t* out_class, int32_t* out_index) {
    if (n_slice < 0 || n_tile < 0 || n_wg <= 0 || mode < 0 || (mode > 4 && mode < 16) || mode > 255 || (n_slice > 0 && !slice_costs) || (n_tile > 0 && !tile_costs) ||
        (n_slice + n_tile > 0 && (!out_class || !out_index))) return HISPMV_EINVAL;
    try {
        const std::vector<double> a(slice_costs, slice_costs + n_slice), b(tile_costs, tile_costs + n_tile);
        const auto order = order_step_queue(a, b, n_wg, mode);
        for (size_t i = 0; i < order.size(); ++i) { out_class[i] = order[i].first; out_index[i] = order[i].second; }
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

// inside[nnz]: 1 for the CSR entries whose block of x lies in the window of their workgroup under the launch plan for n_cus CUs
// (the criterion of the stray split, hispmv_matrix_info.tile_kind 3).  For tests: lets the wavefront model pack the two parts.
HISPMV_API int hispmv_prep_window_membership(const hispmv_prep* p, int n_cus, uint8_t* inside) {
    if (!p || !inside || n_cus <= 0) return HISPMV_EINVAL;
    try {
        SliceStream copy = build_stream(p->csr);
        const LaunchPlan plan = make_plan(copy, n_cus);
        const std::vector<uint8_t> m = window_membership(p->csr, plan);
        std::copy(m.begin(), m.end(), inside);
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}

HISPMV_API int hispmv_prep_dims(const hispmv_prep* p, int64_t d[8]) {
    if (!p || !d) return HISPMV_EINVAL;
    d[0] = p->st.rows; d[1] = p->st.cols; d[2] = p->st.nnz; d[3] = p->st.n_elems; d[4] = p->st.n_slices;
    d[5] = kSliceElems; d[6] = (int64_t)p->st.fix.size(); d[7] = p->st.bytes();
    return HISPMV_OK;
}
HISPMV_API int hispmv_prep_plan(const hispmv_prep* p, int n_cus, int64_t plan[6]) {
    if (!p || !plan || n_cus <= 0) return HISPMV_EINVAL;
    SliceStream copy = p->st;                      // make_plan rewrites the words of staged groups
    const LaunchPlan g = make_plan(copy, n_cus);
    plan[0] = g.block_threads; plan[1] = g.group_slices; plan[2] = g.lds_floats; plan[3] = g.ytile_floats;
    plan[4] = (int64_t)g.groups.size();
    plan[5] = ((int64_t)g.lds_floats + (int64_t)g.ytile_floats * (g.block_threads / 64)) * 4;
    return HISPMV_OK;
}

HISPMV_API int hispmv_prep_apply_plan(hispmv_prep* p, int n_cus, int64_t counts[2]) {
    if (!p || !counts || n_cus <= 0) return HISPMV_EINVAL;
    p->plan = make_plan(p->st, n_cus);
    counts[0] = (int64_t)p->plan.groups.size(); counts[1] = (int64_t)p->plan.frags.size();
    return HISPMV_OK;
}
// The device layout of the stream under the plan hispmv_prep_apply_plan computed (call that first): counts = {bytes, groups,
// compact slices, slices with stray slots, stray-area floats, window floats of the plan}; arrays through hispmv_prep_device_array.
HISPMV_API int hispmv_prep_device_stream(hispmv_prep* p, int64_t counts[6]) {
    if (!p || !counts) return HISPMV_EINVAL;
    try {
        p->dstream = pack_device_stream(p->st, p->plan);
        counts[0] = (int64_t)p->dstream.bytes.size(); counts[1] = (int64_t)p->dstream.groups.size() / 4; counts[2] = p->dstream.compact_slices;
        counts[3] = p->dstream.stray_slices; counts[4] = p->dstream.stray_floats; counts[5] = p->plan.lds_floats;
        return HISPMV_OK;
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
}
// The same layout written by the DEVICE (layout_on_device, what hispmv_load_matrices runs with HISPMV_LAYOUT=device) from the planned host words:
// call hispmv_prep_apply_plan and hispmv_prep_device_stream first; bytes_out takes counts[0] bytes, stray_cols_out n_slices x 64 u32
// (may be null when counts[4] == 0).  For tests: device == host, byte for byte.
HISPMV_API int hispmv_prep_device_stream_on_device(hispmv_prep* p, int device_id, uint8_t* bytes_out, uint32_t* stray_cols_out) {
    if (!p || !bytes_out) return HISPMV_EINVAL;
    if (p->dstream.groups.empty() || (int64_t)p->st.words.size() != p->st.n_slices * kSliceElems) { g_prep_err = "call hispmv_prep_apply_plan and hispmv_prep_device_stream first"; return HISPMV_ESTATE; }
    if (hipSetDevice(device_id) != hipSuccess) { g_prep_err = "no such HIP device"; return HISPMV_EDEVICE; }
    const int64_t ns = p->st.n_slices;
    const size_t wb = p->st.words.size() * sizeof(uint64_t), gb = p->dstream.groups.size() * sizeof(int32_t), sb = (size_t)ns * kStraySlots * sizeof(uint32_t);
    void *dw = nullptr, *dg = nullptr, *db = nullptr, *ds = nullptr;
    auto done = [&](int rc, const char* what) { if (rc != HISPMV_OK) g_prep_err = what; (void)hipFree(dw); (void)hipFree(dg); (void)hipFree(db); (void)hipFree(ds); return rc; };
    if (hipMalloc(&dw, std::max<size_t>(wb, 8)) != hipSuccess || hipMalloc(&dg, gb) != hipSuccess || hipMalloc(&db, std::max<int64_t>(p->dstream.n_bytes, 8)) != hipSuccess)
        return done(HISPMV_EDEVICE, "hipMalloc failed");
    const bool strays = p->dstream.any_stray && stray_cols_out;
    if (strays && (hipMalloc(&ds, sb) != hipSuccess || hipMemset(ds, 0xff, sb) != hipSuccess)) return done(HISPMV_EDEVICE, "hipMalloc failed");
    if (hipMemcpy(dw, p->st.words.data(), wb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dg, p->dstream.groups.data(), gb, hipMemcpyHostToDevice) != hipSuccess)
        return done(HISPMV_EDEVICE, "hipMemcpy failed");
    if (layout_on_device((const uint64_t*)dw, ns, p->plan.group_slices, (const int32_t*)dg, p->plan.lds_floats, p->plan.block_threads / 64, (uint8_t*)db, (uint32_t*)ds, nullptr) != 0)
        return done(HISPMV_EDEVICE, "layout kernel launch failed");
    if (hipDeviceSynchronize() != hipSuccess) return done(HISPMV_EDEVICE, "layout kernel failed");
    if (hipMemcpy(bytes_out, db, (size_t)p->dstream.n_bytes, hipMemcpyDeviceToHost) != hipSuccess) return done(HISPMV_EDEVICE, "hipMemcpy failed");
    if (strays && hipMemcpy(stray_cols_out, ds, sb, hipMemcpyDeviceToHost) != hipSuccess) return done(HISPMV_EDEVICE, "hipMemcpy failed");
    return done(HISPMV_OK, "");
}
HISPMV_API const void* hispmv_prep_device_array(const hispmv_prep* p, int which) {
    if (!p) return nullptr;
    switch (which) {
        case 0: return p->dstream.bytes.data();
        case 1: return p->dstream.groups.data();
        case 2: return p->dstream.stray_cols.data();
        default: return nullptr;
    }
}
HISPMV_API const int32_t* hispmv_prep_groups(const hispmv_prep* p) { return (const int32_t*)p->plan.groups.data(); }
HISPMV_API const int32_t* hispmv_prep_frags(const hispmv_prep* p) { return (const int32_t*)p->plan.frags.data(); }

HISPMV_API int hispmv_prep_build_tts(hispmv_prep* p, int64_t target_tile_elems, int small_geometry, int64_t counts[8], double* lines_per_gather) {
    if (!p || !counts) return HISPMV_EINVAL;
    try {
        TtsGeometry geo;
        if (small_geometry == 1) { geo.max_slots = kTtsSmallSlots; geo.max_rows = kTtsSmallRows; geo.tiles_wanted = 512; }
        if (small_geometry == 6) geo.zero_fill = true;      // the standard sizes without filler words (HISPMV_TTS_GEOMETRY=zerofill)
        if ((small_geometry >= 2 && small_geometry < 6) || small_geometry == 8 || small_geometry == 9) {         // 2 + q / 4 + q / 8 + q: column part q of the tall / paired / tall gap-coded geometry, as the loader builds it for a 256-CU device
            const bool paired = small_geometry >= 4 && small_geometry < 6, gap = small_geometry >= 8;
            const int q = small_geometry - (gap ? 8 : paired ? 4 : 2);
            if (q >= kTtsTallParts) { g_prep_err = "no such column part"; return HISPMV_EINVAL; }
            const std::vector<int32_t> cuts = tts_column_cuts(p->csr, kTtsTallParts);
            const Csr part = csr_column_range(p->csr, q == 0 ? 0 : cuts[(size_t)q - 1], q + 1 == kTtsTallParts ? p->csr.cols : cuts[(size_t)q]);
            p->tts = build_tts(part, target_tile_elems, gap ? tts_tallgap_geometry(256, kTtsTallParts) : paired ? tts_paired_geometry(256) : tts_tall_geometry(256, kTtsTallParts));
        } else
        p->tts = build_tts(p->csr, target_tile_elems, geo);
    } catch (const std::exception& ex) { g_prep_err = ex.what(); return HISPMV_EINVAL; }
    const TtsStream& t = p->tts;
    counts[0] = (int64_t)t.tiles.size(); counts[1] = (int64_t)t.blocks.size(); counts[2] = (int64_t)t.col_base.size();
    counts[3] = (int64_t)t.chunk_info.size() / 2; counts[4] = t.n_fillers; counts[5] = t.n_pad_words; counts[6] = t.max_rows; counts[7] = t.max_slots;
    if (lines_per_gather) *lines_per_gather = t.lines_per_gather;
    return HISPMV_OK;
}
HISPMV_API int hispmv_prep_tts_pieces(const hispmv_prep* p, int64_t counts[2]) {
    if (!p || !counts) return HISPMV_EINVAL;
    counts[0] = (int64_t)p->tts.fix.size() / 4; counts[1] = p->tts.n_carry;
    return HISPMV_OK;
}
HISPMV_API const void* hispmv_prep_tts_array(const hispmv_prep* p, int which) {
    if (!p) return nullptr;
    switch (which) {
        case 6: return p->tts.fix.data();
        case 7: return p->tts.flags_hi.data();
        case 0: return p->tts.words.data();
        case 1: return p->tts.col_base.data();
        case 2: return p->tts.flags.data();
        case 3: return p->tts.chunk_info.data();
        case 4: return p->tts.tiles.data();
        case 5: return p->tts.blocks.data();
        default: return nullptr;
    }
}

HISPMV_API const int64_t* hispmv_prep_csr_row_ptr(const hispmv_prep* p) { return p->csr.row_ptr.data(); }
HISPMV_API const int32_t* hispmv_prep_csr_col(const hispmv_prep* p) { return p->csr.col.data(); }
HISPMV_API const float* hispmv_prep_csr_val(const hispmv_prep* p) { return p->csr.val.data(); }
HISPMV_API const uint64_t* hispmv_prep_words(const hispmv_prep* p) { return p->st.words.data(); }
HISPMV_API const int32_t* hispmv_prep_slice_hdr(const hispmv_prep* p) { return (const int32_t*)p->st.hdr.data(); }
HISPMV_API const int32_t* hispmv_prep_fix(const hispmv_prep* p) { return (const int32_t*)p->st.fix.data(); }
