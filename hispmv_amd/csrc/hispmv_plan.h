// hispmv_plan.h -- per-matrix launch plan of the slice kernel: how many slices a workgroup owns, how many
// wavefronts it has, and which FRAGMENTS of x it stages in LDS.  The MI355X analogue of the reference's
// per-matrix configuration search (automation_tool/src/dse.py:23-95 picks channel counts and window sizes
// per matrix) and of its x window (LoadB fills a BRAM window per column tile, base_functions.cpp:105-150):
// here the "window" of a workgroup is the set of 64-byte blocks of x its slices touch (the most used ones, if
// they do not all fit), staged as a list of contiguous fragments; the element words of such a group carry the
// LDS index instead of the column, or kGlobalColBit | column for the few elements outside the window.
// Host-only code.
#pragma once
#include <cstdint>
#include <vector>

#include "hispmv_prep.h"

namespace hispmv {

constexpr int kFragBlock = 16;            // floats per x block (64 B: one cache-line sector)
constexpr int kFragMaxLen = 2048;         // fragments longer than this are split so that wavefronts share the staging
constexpr int kMaxLdsFloats = kCompactMaxIndex;   // largest x window: 32768 floats (128 KiB), what a compact meta can index

struct GroupDesc {        // 16 B per workgroup
    int32_t frag_begin;   // first fragment of the group in the fragment table
    int32_t frag_count;   // 0: the group gathers x through L2 (its words keep global columns)
    int32_t lds_floats;   // floats of x the group stages
    int32_t n_global;     // elements of the group whose block is NOT staged (their words carry kGlobalColBit | column
                          //   and gather through L2): the window holds the group's most used blocks
};

struct Frag {             // 16 B
    int32_t col_start;    // first column (multiple of kFragBlock)
    int32_t len;          // floats (multiple of kFragBlock, <= kFragMaxLen)
    int32_t lds_off;      // where it lands in the workgroup's window (multiple of kFragBlock)
    int32_t pad;
};

struct LaunchPlan {
    int block_threads = 256;
    int group_slices = 8;
    int lds_floats = 0;        // window floats reserved per workgroup (max over the staged groups); 0 = no LDS window
    int ytile_floats = 64;     // row-total tile per wavefront
    int per_cu = 4;            // resident workgroups per CU the plan was sized for
    std::vector<GroupDesc> groups;
    std::vector<Frag> frags;
    std::vector<uint16_t> slice_spills;  // per slice: how many of its elements lie outside the group's window (empty: none anywhere)
    int64_t staged_floats = 0; // sum over groups (diagnostics)
    int64_t global_elems = 0;  // elements of staged groups that still gather through L2
};

// Chooses the plan for `st` on a device with n_cus compute units and REWRITES the column field of the
// elements of every LDS-staged group in st.words to the element's index in that group's window.
// only_cfg >= 0: evaluate that one configuration of the planner's table only (add_batch_layout: the resident configuration of the part's
// first plan with longer groups -- five of six evaluations saved, 80 -> 15 ms for a matrix of TSOPF's size)
LaunchPlan make_plan(SliceStream& st, int n_cus, int only_cfg = -1);
constexpr int kPlanCfgResident512 = 2, kPlanCfgResident1024 = 5;

// The words of a planned stream with COLUMNS in their meta field again (the inverse of make_plan's rewrite; hispmv_choose.cpp plans a
// part a second time for its batch layout).
WordVec unplanned_words(const SliceStream& st, const LaunchPlan& plan);

// LDS floats one wavefront needs for the row totals of a slice (largest number of rows ending in one slice).
int ytile_floats_for(const SliceStream& st);

// Device form of a planned stream (hispmv_format.h: structure-of-arrays slices, compact or wide per GROUP -- a group is
// compact when it has a window and none of its elements lies outside it).
// STRAY SLOTS (round 4): a group whose slices each have at most kStraySlots elements outside the window stays COMPACT.  The x
// values of a slice's strays are fetched by the wavefront that owns the slice -- one lane per stray, from the slice's list of stray
// columns -- into that wavefront's 64-float stray area behind the window, and the strays' 16-bit metas index that area: stray k of
// the slice at position p of its workgroup's walk sits at window index  lds_floats + (p mod wavefronts) * 64 + k  (the walk is the
// kernel's: slice s of group g is at position (s - rot + n) mod n with rot = 29 g mod n, n = slices of the group).  Without them one
// stray element makes its whole group take 8-byte elements and the two-way gather: the PFlow_742 stand-in falls from 0.70 to 0.43
// of the roofline when 2 % of its entries are re-drawn at random columns (profiles/r4_standin_sweep*.json).
struct DeviceStream {
    std::vector<uint8_t, DefaultInitAllocator<uint8_t>> bytes;   // the slices, group after group (uninitialised until the packing loop writes them)
    std::vector<int32_t> groups;         // n_groups x {frag_begin, frag_count, offset of the group's first slice in kSliceUnit, 1 = compact | 2 = has stray slots}
    std::vector<uint32_t> stray_cols;    // n_slices x kStraySlots columns (0xffffffff = unused), empty when no group uses stray slots
    int stray_floats = 0;                // LDS floats of the stray areas (wavefronts x kStraySlots) behind the window, 0 = none
    int64_t compact_slices = 0, stray_slices = 0;
    int64_t n_bytes = 0;                 // size of `bytes`, also when they are left to the device (materialize = false)
    bool any_stray = false;              // some group uses stray slots (stray_cols has n_slices x kStraySlots entries once materialized)
};
// materialize = false: the group table, sizes and flags only -- the loader then runs layout_on_device (hispmv_prep_device.h) over the
// uploaded host words, which writes the same bytes and stray columns.
DeviceStream pack_device_stream(const SliceStream& st, const LaunchPlan& plan, bool materialize = true);
// Share of the plan's elements outside their windows that stray slots will serve (groups whose slices have <= kStraySlots each).
double stray_slot_coverage(const SliceStream& st, const LaunchPlan& plan);

}  // namespace hispmv
