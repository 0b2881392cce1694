// batch.hpp -- utterance batches resident in HBM (C ABI handle `SRBatch *`) and the tile
// tables that map workgroups onto (utterance, frame range).
#pragma once

#include "common.hpp"

#include <memory>
#include <vector>

namespace sr {

// One scoring workgroup covers `count` (<= 256*F) consecutive frames of ONE utterance, so the
// per-utterance sums never need a segmented reduction inside the kernel.
struct TileDesc {
    int64_t start;  // first frame (global row index)
    int32_t count;  // valid frames in this tile
    int32_t utt;
};

struct TileTable {
    int frames_per_tile = 0;
    int n_tiles = 0;
    DevBuf<TileDesc> d_tiles;
    DevBuf<int> d_utt_tile_begin;  // [U+1]
    std::vector<TileDesc> h_tiles; // host copy (the partial-product path maps noted tiles back to utterances)
    // 32-frame tables: the pipelined shared-sigma kernel's view -- the same tiles followed by its work items (int4 {tile, tile or -1,
    // tile or -1, tile or -1}: a full tile each, the ragged tail tiles of different utterances packed up to four to a wave when
    // `work_packed`), padded with empty items {-1, ..} to whole rounds of 8 workgroups x 12 waves.  Built on first use.
    DevBuf<TileDesc> d_tiles_work;
    int n_work = 0;
    bool work_packed = false;
    // the batch changed its layout: rebuilt on next use INTO the same device buffers (SRBatch::tiles_for).  Dropping the table
    // instead meant two or three hipFree -- each a device synchronisation -- and as many hipMalloc per call of a loop whose
    // utterances differ in length (the reference's predict_one loop, gmmset.py:62-64; sr_batch_reset_pcm; mfcc_extract_batch)
    bool stale = false;
    // (a rebuilt table travels through these, left in flight: common.hpp)
    StagedUpload<TileDesc> stage_tiles, stage_work;
    StagedUpload<int> stage_begin;
};

void ensure_work_table(TileTable &tt, bool pack_tails);   // gmm_score.hip

}  // namespace sr

struct SRBatch {
    enum Kind { PCM16 = 0, PCMF32 = 1, FEATURES = 2 };
    int kind = FEATURES;
    int device = -1;                // the GPU its buffers live on (set when they are first filled)
    int n_utt = 0;
    int dim = 0;                    // features only
    int64_t n_rows = 0;             // samples or frames
    std::vector<int64_t> offsets;   // host copy, [U+1]
    sr::DevBuf<int64_t> d_offsets;
    sr::DevBuf<int16_t> pcm16;
    sr::DevBuf<float> data;         // f32 PCM or features [n_rows][dim]
    std::vector<std::unique_ptr<sr::TileTable>> tile_tables;
    // sr_batch_update_pcm of a small batch (a serving decision's PCM): the samples go through this page-locked copy and the
    // transfer is left in flight on the stream -- the caller's buffer is free when the call returns, the host never waits;
    // `stage_done` says when the staging area may be overwritten by the next update
    sr::PinnedBuf<int16_t> h_stage;
    sr::EventHolder stage_done;
    // a REFILLED batch's offsets, and a refilled feature batch's rows (sr_batch_reset_pcm / _features, mfcc_extract_batch into a
    // batch it has filled before): page-locked, in flight (common.hpp: StagedUpload)
    sr::StagedUpload<int64_t> stage_offsets;
    sr::StagedUpload<float> stage_rows;

    sr::TileTable &tiles_for(int frames_per_tile);
    void invalidate_tiles() {       // after a change of `offsets`
        for (auto &t : tile_tables) t->stale = true;
    }
    // binds an empty batch to the calling thread's device / refuses one that lives elsewhere
    void bind_device() {
        // HIP's current device is per host thread: a thread that chose device d and whose first library call is one that
        // only (re)fills a batch must not allocate on device 0 (ensure_device is what calls hipSetDevice)
        sr::ensure_device();
        if (device < 0) device = sr::current_device();
        if (device != sr::current_device())
            sr::fail("batch lives on device %d, the calling thread is on device %d", device, sr::current_device());
    }
};
