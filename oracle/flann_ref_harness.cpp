// ORACLE SUPPORT - test infrastructure only, never linked into the product.
//
// The reference's matcher (FeatureMatching/AnnMatcher.cpp) gets its neighbours
// from FLANN, which IS vendored under the reference tree and is header-only:
//   /root/reference/cpp/third-party/flann/src/cpp/flann/flann.hpp
// This harness is compiled directly against those headers where they lie
// (oracle/Makefile, target `_ref`; output oracle/_ref/libflann_ref.so) - no
// stand-ins, no copied sources - and calls FLANN exactly as the reference's
// call sites do:
//   flann::Index<flann::L2<float>>(data, params).buildIndex()  AnnMatcher.cpp:227-234
//   tree.knnSearch(query, indices, dists, 3, SearchParams())   AnnMatcher.cpp:122
//   tree.knnSearch(query, indices, dists, 2, SearchParams())   AnnMatcher.cpp:105
//   tree.radiusSearch(query, indices, dists, radius, params)   AnnMatcher.cpp:137
// with the index parameters selectable:
//   kind 0 = flann::LinearIndexParams()   exact search: what pins the
//            exhaustive restatement in oracle/sift_ref.hpp (neighbour order,
//            distance arithmetic, tie order, strict radius) bit for bit;
//   kind 1 = flann::KDTreeIndexParams(8)  the reference's REAL configuration
//            (AnnMatcher.cpp:227; default SearchParams: 32 checks), seeded with
//            flann::seed_random(0): approximate - used to record how far the
//            exact lists are from what the reference returns.
// The match construction around the searches follows append_nearest_neighbors
// and compute_matches (AnnMatcher.cpp:59-268) as a second, independent
// restatement written against FLANN's own containers.
#include <flann/flann.hpp>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace {

  using Index = flann::Index<flann::L2<float>>;

  std::unique_ptr<Index> build(const float* data, int n, int dim, int kind)
  {
    flann::Matrix<float> m(const_cast<float*>(data), size_t(n), size_t(dim));
    std::unique_ptr<Index> idx;
    if (kind == 0)
      idx.reset(new Index(m, flann::LinearIndexParams()));
    else
    {
      flann::seed_random(0);
      idx.reset(new Index(m, flann::KDTreeIndexParams(8)));
    }
    idx->buildIndex();
    return idx;
  }

  struct Match
  {
    int32_t x_index, y_index;
    float score;
    int32_t rank, direction;
  };

  struct Feature  // the OERegion fields KeyProximity / operator== read
  {
    float x, y, m00, m10, m01, m11, orientation, type;
    bool same(const Feature& o) const
    {
      return x == o.x && y == o.y && m00 == o.m00 && m10 == o.m10 &&
             m01 == o.m01 && m11 == o.m11 && orientation == o.orientation &&
             type == o.type;
    }
  };

  // FeatureMatching/KeyProximity.cpp:17-30, Geometry/Tools/Metric.hpp:47-50
  struct Proximity
  {
    float metric2, pixel2;
    static float quad(const Feature& f, float vx, float vy)
    {
      const float r0 = f.m00 * vx + f.m01 * vy;
      const float r1 = f.m10 * vx + f.m11 * vy;
      return vx * r0 + vy * r1;
    }
    bool operator()(const Feature& a, const Feature& b) const
    {
      const float vx = b.x - a.x, vy = b.y - a.y;
      const float sd1 = quad(a, vx, vy), sd2 = quad(b, vx, vy);
      const float dx = a.x - b.x, dy = a.y - b.y;
      return dx * dx + dy * dy < pixel2 || sd1 < metric2 || sd2 < metric2;
    }
  };

  // one direction: every row of `q` against the index built over `t`
  void one_direction(const float* q, int nq, int nt, int dim, Index& tree,
                     float thres2, int direction, bool self,
                     const Proximity& too_close, const Feature* fq,
                     const Feature* ft, size_t max_neighbors,
                     std::vector<Match>& out)
  {
    std::vector<int> vi(std::max<size_t>(max_neighbors, 3));
    std::vector<float> vd(std::max<size_t>(max_neighbors, 3));
    const flann::SearchParams sp;
    auto push = [&](int i1, int i2, float score, int rank) {
      out.push_back(direction == 0 ? Match{i1, i2, score, rank, 0}
                                   : Match{i2, i1, score, rank, 1});
    };
    for (int i = 0; i < nq; ++i)
    {
      flann::Matrix<float> query(const_cast<float*>(q) + size_t(i) * dim, 1,
                                 size_t(dim));
      flann::Matrix<int> indices(vi.data(), 1, max_neighbors);
      flann::Matrix<float> dists(vd.data(), 1, max_neighbors);
      if (nt == 0)
        return;
      if (nt == 1 && !self)
      {
        if (1.f < thres2)
          push(i, 0, 1.f, 1);
        continue;
      }
      if (nt == 2 && self)
      {
        tree.knnSearch(query, indices, dists, 2, sp);
        if (1.f < thres2)
          push(i, indices[0][1], 1.f, 1);
        continue;
      }
      tree.knnSearch(query, indices, dists, 3, sp);
      const int top1 = self ? 1 : 0;
      const float d_top1 = dists[0][top1];
      const float top1_score =
          dists[0][top1 + 1] > 0.f ? d_top1 / dists[0][top1 + 1] : 0.f;
      int K = 1;
      if (thres2 > 1.f)
        K = tree.radiusSearch(query, indices, dists, d_top1 * thres2, sp);
      // after a radius search rank `top1` is read again from the NEW lists,
      // whose first entry's distance is what the scores divide by (:146-147)
      for (int rank = top1; rank < K; ++rank)
      {
        float score = 0.f;
        if (rank == top1)
          score = top1_score;
        else if (dists[0][top1])
          score = dists[0][rank] / dists[0][top1];
        if (score > thres2)
          break;
        const int i2 = indices[0][rank];
        if (self && too_close(fq[i], ft[i2]))
          continue;
        push(i, i2, score, top1 == 0 ? rank + 1 : rank);
      }
    }
  }

  // AnnMatcher.cpp:239-262; equal keys are ordered like the oracle orders them
  // (the reference's std::sort leaves them unspecified)
  void finish(std::vector<Match>& m, const Feature* f1, const Feature* f2)
  {
    std::sort(m.begin(), m.end(), [](const Match& a, const Match& b) {
      if (a.x_index != b.x_index)
        return a.x_index < b.x_index;
      if (a.y_index != b.y_index)
        return a.y_index < b.y_index;
      if (a.score != b.score)
        return a.score < b.score;
      if (a.direction != b.direction)
        return a.direction < b.direction;
      return a.rank < b.rank;
    });
    m.erase(std::unique(m.begin(), m.end(),
                        [&](const Match& a, const Match& b) {
                          if (f1 && f2)
                            return f1[a.x_index].same(f1[b.x_index]) &&
                                   f2[a.y_index].same(f2[b.y_index]);
                          return a.x_index == b.x_index && a.y_index == b.y_index;
                        }),
            m.end());
    std::sort(m.begin(), m.end(), [](const Match& a, const Match& b) {
      if (a.score != b.score)
        return a.score < b.score;
      if (a.x_index != b.x_index)
        return a.x_index < b.x_index;
      return a.y_index < b.y_index;
    });
  }

  int emit(const std::vector<Match>& m, int32_t* out, int capacity)
  {
    const int n = int(m.size());
    for (int i = 0; i < n && i < capacity; ++i)
      std::memcpy(out + 5 * size_t(i), &m[size_t(i)], sizeof(Match));
    return n;
  }

}  // namespace

extern "C" {

//! knnSearch(k) of every query row; idx / dist: nq x k.
int flann_ref_knn(const float* data, int n, int dim, const float* queries, int nq,
                  int k, int kind, int* idx, float* dist)
{
  try
  {
    auto tree = build(data, n, dim, kind);
    const flann::SearchParams sp;
    for (int i = 0; i < nq; ++i)
    {
      flann::Matrix<float> q(const_cast<float*>(queries) + size_t(i) * dim, 1,
                             size_t(dim));
      flann::Matrix<int> I(idx + size_t(i) * k, 1, size_t(k));
      flann::Matrix<float> D(dist + size_t(i) * k, 1, size_t(k));
      tree->knnSearch(q, I, D, size_t(k), sp);
    }
    return 0;
  }
  catch (...)
  {
    return -1;
  }
}

//! radiusSearch(radius[i]) of every query row into rows of `max_nn` entries
//! (the reference passes buffers of max(n1, n2) entries); count[i] = return
//! value.
int flann_ref_radius(const float* data, int n, int dim, const float* queries,
                     int nq, const float* radius, int max_nn, int kind, int* idx,
                     float* dist, int* count)
{
  try
  {
    auto tree = build(data, n, dim, kind);
    const flann::SearchParams sp;
    for (int i = 0; i < nq; ++i)
    {
      flann::Matrix<float> q(const_cast<float*>(queries) + size_t(i) * dim, 1,
                             size_t(dim));
      flann::Matrix<int> I(idx + size_t(i) * max_nn, 1, size_t(max_nn));
      flann::Matrix<float> D(dist + size_t(i) * max_nn, 1, size_t(max_nn));
      count[i] = tree->radiusSearch(q, I, D, radius[i], sp);
    }
    return 0;
  }
  catch (...)
  {
    return -1;
  }
}

//! AnnMatcher{keys1, keys2, ratio}.compute_matches() on FLANN.  matches: rows
//! of (x_index, y_index, score bits, rank, direction).  -> count or -1.
int flann_ref_compute_matches(const float* desc1, int n1, const float* desc2,
                              int n2, int dim, float ratio, int kind,
                              int32_t* matches, int capacity)
{
  try
  {
    if (n1 == 0 || n2 == 0)
      return -1;
    const float thres2 = ratio * ratio;
    auto tree1 = build(desc1, n1, dim, kind);
    auto tree2 = build(desc2, n2, dim, kind);
    const size_t max_nn = size_t(std::max(n1, n2));
    std::vector<Match> m;
    const Proximity unused{0.f, 0.f};
    one_direction(desc1, n1, n2, dim, *tree2, thres2, 0, false, unused, nullptr,
                  nullptr, max_nn, m);
    one_direction(desc2, n2, n1, dim, *tree1, thres2, 1, false, unused, nullptr,
                  nullptr, max_nn, m);
    finish(m, nullptr, nullptr);
    return emit(m, matches, capacity);
  }
  catch (...)
  {
    return -1;
  }
}

//! AnnMatcher{keys, ratio, metric thres, pixel thres}.compute_matches()
//! (self-matching, AnnMatcher.cpp:199-215).  features: n x 8 floats (x, y, m00,
//! m10, m01, m11, orientation, type).
int flann_ref_compute_self_matches(const float* desc, const float* features, int n,
                                   int dim, float ratio, float metric_thres,
                                   float pixel_thres, int kind, int32_t* matches,
                                   int capacity)
{
  try
  {
    if (n == 0)
      return -1;
    const float thres2 = ratio * ratio;
    auto tree1 = build(desc, n, dim, kind);
    auto tree2 = build(desc, n, dim, kind);
    const Feature* f = reinterpret_cast<const Feature*>(features);
    const Proximity too_close{metric_thres * metric_thres,
                              pixel_thres * pixel_thres};
    std::vector<Match> m;
    one_direction(desc, n, n, dim, *tree2, thres2, 0, true, too_close, f, f,
                  size_t(n), m);
    one_direction(desc, n, n, dim, *tree1, thres2, 1, true, too_close, f, f,
                  size_t(n), m);
    finish(m, f, f);
    return emit(m, matches, capacity);
  }
  catch (...)
  {
    return -1;
  }
}

}  // extern "C"
