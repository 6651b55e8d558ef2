// Host-side restatement of the ONE third-party algorithm on the hot path whose tie behaviour is observable:
//   sklearn.neighbors.NearestNeighbors(n_neighbors=k, metric='euclidean').fit(Z2).kneighbors(Z[anchors])    (Model.py:82-86)
// for a 1-column Z -- the label bank, the conditioning variable of the `ta_c` / `tv_c` CMI estimators (Model.py:327,335).
// Real MOSI / MOSEI labels are discrete, so every anchor has many bank rows at distance 0 and WHICH k of them become the
// product sample is decided by scikit-learn's tie order.  With algorithm='auto' and n_features <= 15 scikit-learn builds a
// KDTree (leaf_size 30, float64); this file restates that tree for one feature (scikit-learn 1.7.2, the version in this image):
//   build : sklearn/neighbors/_binary_tree.pxi.tp:876-878 (n_levels, n_nodes), :1035-1085 (_recursive_build),
//           sklearn/neighbors/_partition_nodes.pyx:24-60 (std::nth_element under the (value, index) comparator)
//   query : _binary_tree.pxi.tp:1606-1660 (_query_single_depthfirst), sklearn/neighbors/_kd_tree.pyx.tp:123-147 (min_rdist),
//           sklearn/utils/_heap.pyx (heap_push: strict improvement only), sklearn/utils/_sorting.pyx (simultaneous_sort)
// The device kernel (estimator_ops.hip: knn_kernel) breaks ties towards the lower bank row instead -- an equally valid
// kNN; the host-anchor mode of Model / Solver (the mode that replays the reference's numpy anchor draws) uses THIS routine for
// the two label-conditioned calls so that the product sample is the reference's, row for row.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "common.h"

namespace {

struct Tree {
  const double* data;
  std::vector<long> idx;
  std::vector<long> start, end;
  std::vector<char> leaf;
  std::vector<double> lo, hi;
  long n_nodes = 0;
  int leaf_size = 30;

  void build(long i_node, long s, long e) {
    double mn = std::numeric_limits<double>::infinity(), mx = -mn;
    for (long i = s; i < e; ++i) { const double v = data[idx[i]]; mn = std::min(mn, v); mx = std::max(mx, v); }
    lo[i_node] = mn; hi[i_node] = mx; start[i_node] = s; end[i_node] = e;
    if (2 * i_node + 1 >= n_nodes || e - s < 2) { leaf[i_node] = 1; return; }
    leaf[i_node] = 0;
    const long n_mid = (e - s) / 2;
    const double* d = data;
    std::nth_element(idx.begin() + s, idx.begin() + s + n_mid, idx.begin() + e,
                     [d](long a, long b) { return d[a] == d[b] ? a < b : d[a] < d[b]; });
    build(2 * i_node + 1, s, s + n_mid);
    build(2 * i_node + 2, s + n_mid, e);
  }
  double min_rdist(long i_node, double pt) const {
    const double d_lo = lo[i_node] - pt, d_hi = pt - hi[i_node];
    const double d = (d_lo + std::fabs(d_lo)) + (d_hi + std::fabs(d_hi));
    return std::pow(0.5 * d, 2.0);
  }
};

void heap_push(double* values, long* indices, long size, double val, long val_idx) {
  if (val >= values[0]) return;
  values[0] = val; indices[0] = val_idx;
  long cur = 0;
  for (;;) {
    const long l = 2 * cur + 1, r = l + 1;
    long swap;
    if (l >= size) break;
    else if (r >= size) { if (values[l] > val) swap = l; else break; }
    else if (values[l] >= values[r]) { if (val < values[l]) swap = l; else break; }
    else { if (val < values[r]) swap = r; else break; }
    values[cur] = values[swap]; indices[cur] = indices[swap];
    cur = swap;
  }
  values[cur] = val; indices[cur] = val_idx;
}

void dual_swap(double* v, long* i, long a, long b) { std::swap(v[a], v[b]); std::swap(i[a], i[b]); }
void simultaneous_sort(double* v, long* ix, long size) {
  if (size <= 1) return;
  if (size == 2) { if (v[0] > v[1]) dual_swap(v, ix, 0, 1); return; }
  if (size == 3) {
    if (v[0] > v[1]) dual_swap(v, ix, 0, 1);
    if (v[1] > v[2]) { dual_swap(v, ix, 1, 2); if (v[0] > v[1]) dual_swap(v, ix, 0, 1); }
    return;
  }
  long pivot = size / 2;
  if (v[0] > v[size - 1]) dual_swap(v, ix, 0, size - 1);
  if (v[size - 1] > v[pivot]) { dual_swap(v, ix, size - 1, pivot); if (v[0] > v[size - 1]) dual_swap(v, ix, 0, size - 1); }
  const double pv = v[size - 1];
  long store = 0;
  for (long i = 0; i < size - 1; ++i)
    if (v[i] < pv) { dual_swap(v, ix, i, store); ++store; }
  dual_swap(v, ix, store, size - 1);
  pivot = store;
  if (pivot > 1) simultaneous_sort(v, ix, pivot);
  if (pivot + 2 < size) simultaneous_sort(v + pivot + 1, ix + pivot + 1, size - pivot - 1);
}

void query(const Tree& t, long i_node, double pt, double* hv, long* hi_, long k, double lb) {
  if (lb > hv[0]) return;
  if (t.leaf[i_node]) {
    for (long i = t.start[i_node]; i < t.end[i_node]; ++i) {
      const double df = pt - t.data[t.idx[i]];
      heap_push(hv, hi_, k, df * df, t.idx[i]);
    }
    return;
  }
  const long i1 = 2 * i_node + 1, i2 = i1 + 1;
  const double lb1 = t.min_rdist(i1, pt), lb2 = t.min_rdist(i2, pt);
  if (lb1 <= lb2) { query(t, i1, pt, hv, hi_, k, lb1); query(t, i2, pt, hv, hi_, k, lb2); }
  else { query(t, i2, pt, hv, hi_, k, lb2); query(t, i1, pt, hv, hi_, k, lb1); }
}

}  // namespace

extern "C" int mimrl_knn_r1_host(const float* z, int N, const int32_t* anchors, int m, int k, int32_t* idx_out) {
  using namespace mimrl;
  if (!z || !anchors || !idx_out || N <= 0 || m <= 0 || k <= 0) return set_error(MIMRL_ERR_ARG, "knn_r1_host: bad argument");
  std::vector<char> is_anchor(N, 0);
  for (int i = 0; i < m; ++i) {
    if (anchors[i] < 0 || anchors[i] >= N) return set_error(MIMRL_ERR_ARG, "knn_r1_host: anchor %d outside the bank", anchors[i]);
    is_anchor[anchors[i]] = 1;
  }
  std::vector<int32_t> cand;                 // Model.py:83-84: rows of the bank that are not anchors, in bank order
  std::vector<double> data;
  cand.reserve(N); data.reserve(N);
  for (int r = 0; r < N; ++r)
    if (!is_anchor[r]) { cand.push_back(r); data.push_back((double)z[r]); }
  const long n = (long)cand.size();
  // sklearn/neighbors/_base.py (_fit, algorithm='auto'): brute force when n_neighbors >= n_samples // 2 -- a different tie order,
  // not restated (banks that small do not occur: the caller falls back to the device kernel)
  if (k >= n / 2) return set_error(MIMRL_ERR_ARG, "knn_r1_host: bank of %ld candidate rows is in scikit-learn's brute-force regime (k=%d)", n, k);
  Tree t;
  t.data = data.data();
  const int n_levels = (int)(std::log2(std::fmax(1.0, (double)(n - 1) / t.leaf_size)) + 1);
  t.n_nodes = (1L << n_levels) - 1;
  t.idx.resize(n);
  for (long i = 0; i < n; ++i) t.idx[i] = i;
  t.start.assign(t.n_nodes, 0); t.end.assign(t.n_nodes, 0); t.leaf.assign(t.n_nodes, 1);
  t.lo.assign(t.n_nodes, 0.0); t.hi.assign(t.n_nodes, 0.0);
  t.build(0, 0, n);
  std::vector<double> hv(k);
  std::vector<long> hi_(k);
  for (int i = 0; i < m; ++i) {
    const double pt = (double)z[anchors[i]];
    std::fill(hv.begin(), hv.end(), std::numeric_limits<double>::infinity());
    std::fill(hi_.begin(), hi_.end(), 0L);
    query(t, 0, pt, hv.data(), hi_.data(), k, t.min_rdist(0, pt));
    simultaneous_sort(hv.data(), hi_.data(), k);
    for (int q = 0; q < k; ++q) idx_out[(long)i * k + q] = cand[hi_[q]];
  }
  return MIMRL_OK;
}
