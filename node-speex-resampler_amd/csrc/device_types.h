// device_types.h -- plain structs shared by the host engine and the HIP kernels.
#pragma once
#include <cstdint>

#include "diag.h"

namespace speexhip {

// One stream's share of one processing call.  V = history ++ input is the virtual frame
// sequence the FIR windows index (history = the last taps-1 consumed frames, zeros at start:
// reference resample.c:721-725, 898-899; `consumed` counts frames of V past the history, so it
// includes pending frames drained by this call).
struct StreamDesc {
  const void *in;       // interleaved s16 or f32 (the call's sample type), in_frames frames
                        // (device); NULL = silence
  const float *hist;    // hist_frames frames, interleaved, always float like the reference's `mem`
  void *out;            // interleaved s16 or f32, room for n_out frames (device)
  float *hist_next;     // where this call leaves the next call's history
  uint32_t in_frames;
  uint32_t n_out;       // frames to produce (host planner)
  uint32_t consumed;    // input frames entering the history
  int32_t last0;        // window start of output 0, in V-frames minus (taps-1) offset: see pos()
  uint32_t frac0;       // phase numerator of output 0
  uint32_t k_shift;     // phase_index_of(frac0): canonical phase index of output 0
  int32_t base_shift;   // last0 - (k_shift*num) div den: V-frame where period 0, phase 0 starts
  uint32_t tile_begin;  // first tile of this stream in the launch's flat tile list
  uint32_t hist_frames; // frames in `hist`: taps-1, plus the pending ("magic") frames a filter
                        // change left buffered (reference resample.c:727-782) -- they are input
                        // that is already in the line, so V simply starts with a longer history
  uint32_t hist_keep;   // frames this call leaves in hist_next: taps-1 + pending frames remaining
  uint32_t m_total;     // periods this call touches: ceil((k_shift + n_out) / den) (period kernel; a division
                        // every workgroup used to make in its prologue)
};

// A launch carries up to 32 streams (BASELINE configs[4]'s share of one GPU) and their descriptors travel in its
// kernel-argument segment: 2.5 KB of the 4 KB a launch may pass -- no copy in front of the launch, no dependent load in
// the kernels.  Larger batches run as one launch per 32 streams (engine.cpp, run_plans; until round 4 they were one
// launch through a pinned -> device descriptor ring, for which every kernel existed in a second form).
static const int kMaxPackedStreams = 32;
struct DescPack {  // the descriptors of one launch, in its kernel-argument segment
  StreamDesc d[kMaxPackedStreams];
};

struct ExactParams {
  const float *table;   // reference-layout sinc table (device)
  uint32_t table_len;
  uint32_t num, den;
  uint32_t taps, oversample;
  uint32_t channels;
  uint32_t outs_per_block;  // output frames per workgroup (== blockDim.x)
  uint32_t span_cap;        // frames of LDS sample window per workgroup
  // Samples between two frames of one channel in the input / output / history buffers: all equal
  // to `channels` for interleaved calls.  The per-channel entry points (reference
  // resample.c:927-1036 with the strides of :1170-1188) run ONE channel (channels == 1 here)
  // through buffers with strides of their own.
  uint32_t in_stride, out_stride, hist_stride;
  uint32_t zero;            // != 0: resampler_basic_zero (resample.c:561-591): outputs are zeros, the
                            // counters and the history move as usual
};

struct SlideParams {
  const float *rows;        // taps [step][phase], phase rows shifted by delta_r (device, scalar loads)
  uint32_t den, taps, channels;
  uint32_t row_len;         // steps per phase row (taps + largest shift, rounded to the iteration)
  uint32_t cgroups;         // lanes per lane block (channel pairs, or channels for phase pairing)
  uint32_t blocks_per_wave; // lane blocks (P periods each) per wave
  uint32_t blocks_per_tile; // ... per workgroup
  uint32_t row_stride;      // floats between LDS rows (P frames + bank padding)
  uint32_t row_magic;       // ceil(2^32 / floats per row): division-free row index while staging
  uint32_t threads;         // lanes per workgroup (a kernel argument: blockDim.x would be fetched from the
                            // dispatch packet with a vector load that drains the staging loads in flight)
#ifdef SPEEXHIP_DIAG
  uint32_t skip;            // phase-skipping mask (diag.h): the diagnostics build only
#endif
  uint32_t base_waves;      // waves that carry lane blocks (= threads / 64 / parts)
  uint32_t parts;           // > 1: tap-range parts, sets of base_waves waves each a range of the iterations
};

struct PeriodParams {
  const float *rows;      // effective taps, layout [group][s][i] (device, read by scalar loads)
  const uint32_t *delta;  // per group: (g*R*num) div den, first-input offset of the group
  uint32_t l4;            // row length / 4
  uint32_t groups;        // phase groups (R phases each)
  uint32_t num, den, taps, channels;
  uint32_t cgroups;       // channel groups per frame (CT channels each)
  uint32_t lane_periods;  // output periods per workgroup tile
  uint32_t half_periods;  // odd channel counts: lanes carry periods pl and pl + half_periods (0 = one period)
  uint32_t half_offset;   // ... floats between the two periods' samples in the LDS window
  uint32_t wave_groups;   // waves per workgroup; wave w of split z takes groups z*wave_groups+w, ...
  uint32_t tail_frames;   // input frames a period needs beyond its start
  uint32_t history_block; // one-shot form: blockIdx.x of the workgroup that rolls the history
  uint32_t pad;           // LDS bank padding: floats inserted after every period of the window
  uint32_t wrap_step;     // iterations between two period boundaries of a group's window (num/4)
  uint32_t period_magic;  // ceil(2^32 / (num*channels)): division-free period index in the padded image
  uint32_t threads;       // lanes per workgroup (see SlideParams::threads)
  uint32_t prio;          // bit 0: prologue + staging at raised wave priority; bit 1: the stores too
#ifdef SPEEXHIP_DIAG
  uint32_t skip;          // phase-skipping mask (SPEEXHIP_SKIP, diag.h): the diagnostics build only
#endif
  uint32_t ksplit;        // > 1: tap-range shares, this many waves per phase group (fir_tile_parts)
  uint32_t touch;         // != 0: every workgroup fetches the tap rows into L2 beside its window (touch_rows)
};

}  // namespace speexhip
