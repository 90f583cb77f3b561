"""NumPy stand-in for lars_d_quotient_select_hist (tests only): what one rank's kernel pass counts, from index planes."""
import numpy as np

from lars_image_processing_amd import batch


def select_pass_on_planes(planes, sample_planes=None):
    """planes = [NDVI values, GNDVI values] of this rank -> pass_fn(first, buckets[4]) like TileBatch.select_histogram.
    ``sample_planes``: what the subsampled bucket pass (first == 3) sees instead of every 16th value -- a sample from the
    wrong place makes the window miss and the caller fall back."""
    planes = [np.asarray(p, dtype=np.float32) for p in planes]
    pos = [batch.select_position(p) for p in planes]
    spos = [batch.select_position(np.asarray(p, dtype=np.float32)) for p in sample_planes] if sample_planes is not None else None

    def pass_fn(first, buckets):
        out = np.zeros((2, 2, batch.SELECT_BINS), dtype=np.uint64)
        if first == 3:                                       # bucket pass over a subsample, under track 0
            for s in range(2):
                bucket = spos[s][0] if spos is not None else pos[s][0][::16]
                out[s, 0] = np.bincount(bucket, minlength=batch.SELECT_BINS)
            return out
        if first == 2:                                       # window pass: below words | slots | above words, under track 0
            for s in range(2):
                w = batch.select_window_word(planes[s], int(np.int32(np.uint32(int(buckets[2 * s]) & 0xFFFFFFFF))))
                lane = np.arange(w.size) % 64
                w = np.where(w < 64, lane, np.where(w >= 64 + batch.WINDOW_SLOTS, 64 + batch.WINDOW_SLOTS + lane, w))
                out[s, 0] = np.bincount(w, minlength=batch.SELECT_BINS)
            return out
        shared = all(buckets[2 * q] == buckets[2 * q + 1] for q in range(2))
        for s in range(2):
            bucket, slot = pos[s]
            for t in range(2):
                if first:
                    if t == 0:                               # the bucket pass counts under track 0
                        out[s, 0] = np.bincount(bucket, minlength=batch.SELECT_BINS)
                elif t == 0 or not shared:                   # both streams' tracks shared: only track 0 is counted
                    out[s, t] = np.bincount(slot[bucket == int(buckets[s * 2 + t])], minlength=batch.SELECT_BINS)
        return out
    return pass_fn
