// Counter-based dropout masks for the text tower's training-mode dropout.
//
// The reference runs Hugging Face BertModel under model.train() (mmgclip/experiments/ClassifierExperiment.py:97 ->
// mmgclip/networks/encoder.py:156), i.e. with hidden_dropout_prob = attention_probs_dropout_prob = 0.1
// (notebooks/bert_experimental.ipynb:609-624).  torch's Philox stream cannot be reproduced, so the mask is a pure function of
// (seed, site, element index): the backward regenerates it instead of storing it, and oracle/dropout_oracle.py restates it in
// numpy, which is what the parity tests compare against.
//     key  = fmix32(site ^ seed_hi) after adding seed_lo          (one per dropout site and step)
//     keep = fmix32(index * 0x9E3779B1 + key) >= floor(p * 2^32)  (murmur3 finaliser: a bijection of the 32-bit index)
// Element index (independent of the packed / padded token layout):
//     hidden sites   : token * C + column,        token = b * S + s of the PADDED batch
//     attention site : query * 512 + key (positions inside the sequence; S <= 512) under the per-(sequence, head) key
//                      key_bh = fmix32(key + (b * heads + h) * 0xB5297A4D)   - the sequence-head number never enters the 32-bit
//                      index, so batches of any size draw distinct masks per (b, h) (round 2 indexed ((bh * 512 + q) * 512 + k,
//                      which wrapped at bh = 16384 = 1366 sequences of 12 heads)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MMG_HD __host__ __device__ __forceinline__
#else
#define MMG_HD static inline
#endif

MMG_HD unsigned mmg_fmix32(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
MMG_HD unsigned mmg_drop_key(unsigned long long seed, unsigned site) {
    return mmg_fmix32((site ^ (unsigned)(seed >> 32)) + (unsigned)seed);
}
MMG_HD unsigned mmg_drop_key_bh(unsigned key, unsigned bh) { return mmg_fmix32(key + bh * 0xB5297A4Du); }
MMG_HD unsigned mmg_drop_bits(unsigned index, unsigned key) { return mmg_fmix32(index * 0x9E3779B1u + key); }
MMG_HD unsigned mmg_drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (unsigned)t);
}

struct DropArgs {
    unsigned key;        // mmg_drop_key(seed, site)
    unsigned thresh;     // keep iff bits >= thresh ; 0 = dropout off
    float scale;         // 1 / (1 - p)
};
