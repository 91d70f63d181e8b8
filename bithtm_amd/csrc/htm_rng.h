// Keyed random draws on the device.  Twin of oracle/keyed_rng.py -- keep the two in step.
//
//   draw24(base, a, b) -> m in [0, 2^24);  u = m * 2^-24
//   base = stream_base(seed, stream, step)
//
//   stream 1  least-used-cell jitter   a = flat cell id              (networks.py:87)
//   stream 2  synapse-growth priority  a = segment id, b = flat presynaptic cell id
//                                                                    (projections.py:120)
//   stream 3  matching-segment jitter  a = segment id                (projections.py:235)
#pragma once
#include <stdint.h>

#define HTM_STREAM_LEAST_USED 1u
#define HTM_STREAM_GROWTH 2u
#define HTM_STREAM_SEGMENT_JITTER 3u
#define HTM_STREAM_POPULATE_CELL 4u      // pre-populated pools (htm_populate): a = segment id, b = synapse index
#define HTM_STREAM_POPULATE_PERM 5u

__host__ __device__ __forceinline__ uint32_t htm_mix32(uint32_t x) {   // "lowbias32"
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ __forceinline__ uint32_t htm_stream_base(uint32_t seed, uint32_t stream, uint32_t step) {
    uint32_t h = htm_mix32(seed + stream * 0x9E3779B9u);
    return htm_mix32(h ^ step);
}

__host__ __device__ __forceinline__ uint32_t htm_draw24(uint32_t base, uint32_t a, uint32_t b) {
    return htm_mix32(htm_mix32(base ^ a) ^ b) >> 8;
}

// the same hash, all 32 bits (htm_populate: presynaptic cell = (draw32 * cells) >> 32)
__host__ __device__ __forceinline__ uint32_t htm_draw32(uint32_t base, uint32_t a, uint32_t b) {
    return htm_mix32(htm_mix32(base ^ a) ^ b);
}

// float32(float64(count) + u): the in-place `float32 += float64` of networks.py:87 and
// projections.py:235 (computed in float64, rounded once).
__device__ __forceinline__ float htm_jitter(float base_value, uint32_t m24) {
    return (float)((double)base_value + (double)m24 * (1.0 / 16777216.0));
}
