// pm_pairwise.h — the order in which NumPy adds up a contiguous float64 vector, restated so that a sum taken on the device has
// NumPy's bits.  Why: get_similar_transform (reference find_transform.py:43-53) builds its 4 x 4 quaternion matrix from nine
// np.sum(...) of product vectors and takes ROW 0 of np.linalg.eig's eigenvector matrix (:60-66), so the result hangs on the
// last bit of those sums: only NumPy's own summation order reproduces the reference.
//
// np.add.reduce over n contiguous doubles (numpy 2.2.6; numpy/_core/src/umath/loops_utils.h.src: DOUBLE_pairwise_sum, driven by
// the ufunc machinery in buffer-sized pieces):
//   the vector is cut into CHUNKS of 8 192 elements (np.getbufsize(), the default), the chunk sums are added one after the
//   other, first to last;
//   a chunk is summed PAIRWISE: a piece of more than 128 elements is split at half = (len / 2) rounded down to a multiple of 8
//   and its sum is sum(left) + sum(right); a piece of 8..128 elements (a LEAF) runs eight interleaved partial sums
//   r[j] = a[j] + a[8 + j] + a[16 + j] + ..., combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), then adds the
//   len % 8 trailing elements one by one; fewer than 8 elements are added one by one starting from 0.0.
// Checked against np.sum / np.mean for thousands of lengths by tests/test_pairwise_host.py (this header compiled with gcc).
#pragma once

#ifndef PM_HD
#if defined(__HIPCC__)
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD static inline
#endif
#endif

#define PM_PW_CHUNK 8192      // np.getbufsize()
#define PM_PW_LEAF 128        // PW_BLOCKSIZE
#define PM_PW_MAX_DEPTH 16    // a chunk of 8 192 splits 6 times; slack for safety

// Leaves of the vector [0, n), left to right: off[k] .. off[k + 1].  chunk_first[c] = index of the first leaf of chunk c
// (chunk_first[chunks] = number of leaves).  Returns the number of leaves (<= n / 57 + chunks + 1); -1 if cap is too small.
PM_HD int pm_pw_plan(int n, int *off, int cap, int *chunk_first, int chunk_cap) {
    int leaves = 0, chunks = 0;
    for (int c0 = 0; c0 < n; c0 += PM_PW_CHUNK) {
        if (chunks >= chunk_cap - 1) return -1;
        chunk_first[chunks++] = leaves;
        const int clen = (n - c0 < PM_PW_CHUNK) ? n - c0 : PM_PW_CHUNK;
        int st_off[PM_PW_MAX_DEPTH + 2], st_len[PM_PW_MAX_DEPTH + 2], sp = 0;
        st_off[0] = c0; st_len[0] = clen;
        while (sp >= 0) {                           // depth-first, left before right
            const int o = st_off[sp], l = st_len[sp];
            --sp;
            if (l <= PM_PW_LEAF) {
                if (leaves >= cap - 1) return -1;
                off[leaves++] = o;
            } else {
                int half = l / 2;
                half -= half % 8;
                st_off[++sp] = o + half; st_len[sp] = l - half;     // right (popped second)
                st_off[++sp] = o; st_len[sp] = half;                 // left
            }
        }
    }
    off[leaves] = n;
    chunk_first[chunks] = leaves;
    return leaves;
}

PM_HD int pm_pw_leaf_cap(int n) { return n / 56 + n / PM_PW_CHUNK + 8; }
PM_HD int pm_pw_chunk_cap(int n) { return n / PM_PW_CHUNK + 3; }

// One leaf (len <= 128) the way NumPy adds it.
PM_HD double pm_pw_leaf_sum(const double *a, int len) {
    if (len < 8) {
        double res = 0.0;
        for (int i = 0; i < len; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < len - (len % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < len; ++i) res += a[i];
    return res;
}

// The leaf sums of one vector (leaf[k] for the plan's leaves) -> the total, in NumPy's order: inside a chunk the pairwise tree
// (rebuilt from the leaf offsets: a piece [o, o + l) is a leaf iff l <= 128), across chunks first to last.
PM_HD double pm_pw_combine(const double *leaf, const int *off, const int *chunk_first, int chunks, int n) {
    double total = 0.0;
    for (int c = 0; c < chunks; ++c) {
        const int c0 = c * PM_PW_CHUNK;
        const int clen = (n - c0 < PM_PW_CHUNK) ? n - c0 : PM_PW_CHUNK;
        int next = chunk_first[c];
        // post-order evaluation with an explicit stack: frames (len, phase, left value)
        int f_len[PM_PW_MAX_DEPTH + 2], f_phase[PM_PW_MAX_DEPTH + 2], sp = 0;
        double f_left[PM_PW_MAX_DEPTH + 2], ret = 0.0;
        f_len[0] = clen; f_phase[0] = 0;
        while (sp >= 0) {
            const int l = f_len[sp];
            if (l <= PM_PW_LEAF) { ret = leaf[next++]; --sp; continue; }
            int half = l / 2;
            half -= half % 8;
            if (f_phase[sp] == 0) { f_phase[sp] = 1; ++sp; f_len[sp] = half; f_phase[sp] = 0; }
            else if (f_phase[sp] == 1) { f_left[sp] = ret; f_phase[sp] = 2; ++sp; f_len[sp] = l - half; f_phase[sp] = 0; }
            else { ret = f_left[sp] + ret; --sp; }
        }
        (void)off;
        total = (c == 0) ? ret : total + ret;
    }
    return total;
}
