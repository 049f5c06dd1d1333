#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>

namespace mirt {

namespace {

constexpr int kBins = 16;

struct Box {
  float mn[3], mx[3];
  void reset() {
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::numeric_limits<float>::infinity();
      mx[a] = -std::numeric_limits<float>::infinity();
    }
  }
  void grow(const float* bmn, const float* bmx) {
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], bmn[a]);
      mx[a] = std::max(mx[a], bmx[a]);
    }
  }
  float half_area() const {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Child {
  uint32_t ref;
  Box box;
  uint32_t height;
};

uint32_t ceil_log2(size_t n) {
  uint32_t l = 0;
  size_t v = 1;
  while (v < n) { v <<= 1; ++l; }
  return l;
}

struct Builder {
  std::vector<BuildPrim>& prims;
  std::vector<HostNode>& nodes;
  uint32_t leaf_max;
  float step_cost;  // cost of one more step down the tree, in primitive tests

  Child make_leaf(size_t lo, size_t hi) const {
    Child c;
    c.ref = 0x80000000u | ((uint32_t)(hi - lo - 1) << 28) | (uint32_t)lo;
    c.box.reset();
    for (size_t i = lo; i < hi; ++i) c.box.grow(prims[i].mn, prims[i].mx);
    c.height = 0;
    return c;
  }

  float centroid(const BuildPrim& p, int axis) const { return 0.5f * (p.mn[axis] + p.mx[axis]); }

  Child build(size_t lo, size_t hi, uint32_t budget) {
    if (hi - lo == 1) return make_leaf(lo, hi);
    const size_t n = hi - lo;

    Box cb;  // centroid bounds
    Box all;
    cb.reset();
    all.reset();
    for (size_t i = lo; i < hi; ++i) {
      const float c[3] = {centroid(prims[i], 0), centroid(prims[i], 1), centroid(prims[i], 2)};
      cb.grow(c, c);
      all.grow(prims[i].mn, prims[i].mx);
    }
    size_t mid = lo + n / 2;
    const bool force_median = ceil_log2(n) + 1 >= budget;
    bool split_done = false;
    if (!force_median && n >= 2) {
      // binned SAH over the three axes
      float best_cost = std::numeric_limits<float>::infinity();
      int best_axis = -1, best_bin = -1;
      for (int axis = 0; axis < 3; ++axis) {
        const float extent = cb.mx[axis] - cb.mn[axis];
        if (!(extent > 0.0f)) continue;
        Box bin_box[kBins];
        size_t bin_n[kBins];
        for (int b = 0; b < kBins; ++b) { bin_box[b].reset(); bin_n[b] = 0; }
        const float scale = (float)kBins / extent;
        for (size_t i = lo; i < hi; ++i) {
          int b = (int)((centroid(prims[i], axis) - cb.mn[axis]) * scale);
          b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
          bin_box[b].grow(prims[i].mn, prims[i].mx);
          ++bin_n[b];
        }
        float right_area[kBins];
        size_t right_n[kBins];
        Box acc;
        acc.reset();
        size_t cnt = 0;
        for (int b = kBins - 1; b > 0; --b) {
          if (bin_n[b]) acc.grow(bin_box[b].mn, bin_box[b].mx);
          cnt += bin_n[b];
          right_area[b] = acc.half_area();
          right_n[b] = cnt;
        }
        acc.reset();
        cnt = 0;
        for (int b = 0; b < kBins - 1; ++b) {
          if (bin_n[b]) acc.grow(bin_box[b].mn, bin_box[b].mx);
          cnt += bin_n[b];
          if (cnt == 0 || right_n[b + 1] == 0) continue;
          const float cost = acc.half_area() * (float)cnt + right_area[b + 1] * (float)right_n[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
      // a small group stays one leaf when splitting it would not pay: one more node step against the
      // primitive tests the children's smaller boxes would save (a node step costs about one test)
      if (n <= leaf_max && best_axis >= 0 && !(best_cost / std::max(all.half_area(), 1e-30f) + step_cost < (float)n))
        return make_leaf(lo, hi);
      if (best_axis >= 0) {
        const float extent = cb.mx[best_axis] - cb.mn[best_axis];
        const float scale = (float)kBins / extent;
        const float base = cb.mn[best_axis];
        auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const BuildPrim& p) {
          int b = (int)((0.5f * (p.mn[best_axis] + p.mx[best_axis]) - base) * scale);
          b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
          return b <= best_bin;
        });
        mid = (size_t)(it - prims.begin());
        split_done = mid > lo && mid < hi;
      }
    }
    if (!split_done && n <= leaf_max) return make_leaf(lo, hi);  // nothing separates them
    if (!split_done) {
      // balanced split along the widest centroid axis
      int axis = 0;
      float widest = -1.0f;
      for (int a = 0; a < 3; ++a) {
        const float e = cb.mx[a] - cb.mn[a];
        if (e > widest) { widest = e; axis = a; }
      }
      mid = lo + n / 2;
      std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                       [&](const BuildPrim& a, const BuildPrim& b) { return centroid(a, axis) < centroid(b, axis); });
    }
    const uint32_t node_index = (uint32_t)nodes.size();
    nodes.emplace_back();
    const Child l = build(lo, mid, budget - 1);
    const Child r = build(mid, hi, budget - 1);
    HostNode& nd = nodes[node_index];
    for (int a = 0; a < 3; ++a) {
      nd.lmin[a] = l.box.mn[a]; nd.lmax[a] = l.box.mx[a];
      nd.rmin[a] = r.box.mn[a]; nd.rmax[a] = r.box.mx[a];
    }
    nd.left = l.ref;
    nd.right = r.ref;
    nd.lgrow = 0.0f;
    nd.rgrow = 0.0f;
    Child c;
    c.ref = node_index;
    c.box = l.box;
    c.box.grow(r.box.mn, r.box.mx);
    c.height = 1 + std::max(l.height, r.height);
    return c;
  }
};

}  // namespace

BuiltBvh build_bvh(std::vector<BuildPrim>& prims, uint32_t max_height, uint32_t leaf_max, float step_cost) {
  BuiltBvh out;
  if (prims.empty()) return out;
  out.nodes.reserve(prims.size());
  if (const char* e = std::getenv("MIRT_STEP_COST")) step_cost *= (float)std::atof(e);  // tuning aid
  Builder b{prims, out.nodes, std::max(1u, std::min(leaf_max, 8u)), step_cost};
  const Child root = b.build(0, prims.size(), max_height + 1);
  out.root_ref = root.ref;
  out.height = root.height;
  for (int a = 0; a < 3; ++a) { out.mn[a] = root.box.mn[a]; out.mx[a] = root.box.mx[a]; }
  return out;
}

namespace {

struct Refit {
  std::vector<HostNode>& nodes;
  const std::vector<BuildPrim>& prims;
  // box of everything under `ref`; a missing child has the empty box
  Box fit(uint32_t ref) {
    Box b;
    b.reset();
    if (ref == 0xFFFFFFFFu) return b;
    if (ref & 0x80000000u) {
      const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
      for (uint32_t k = first; k < first + count && k < prims.size(); ++k) b.grow(prims[k].mn, prims[k].mx);
      return b;
    }
    const uint32_t left = nodes[ref].left, right = nodes[ref].right;
    const Box l = fit(left), r = fit(right);
    HostNode& nd = nodes[ref];
    for (int a = 0; a < 3; ++a) {
      if (left != 0xFFFFFFFFu) { nd.lmin[a] = l.mn[a]; nd.lmax[a] = l.mx[a]; }
      if (right != 0xFFFFFFFFu) { nd.rmin[a] = r.mn[a]; nd.rmax[a] = r.mx[a]; }
    }
    b = l;
    b.grow(r.mn, r.mx);
    return b;
  }
};

}  // namespace

namespace {

struct Widen {
  const std::vector<HostNode>& in;
  std::vector<WideNode>& out;
  struct Kid {
    uint32_t ref;
    Box box;
  };
  static bool inner(uint32_t ref) { return !(ref & 0x80000000u); }
  // returns the reference that stands for `ref` in the wide tree, and the height below it
  uint32_t convert(uint32_t ref, uint32_t& height) {
    height = 0;
    if (!inner(ref)) return ref;
    Kid kids[4];
    int n = 0;
    auto take = [&](const HostNode& nd) {
      if (nd.left != 0xFFFFFFFFu) {
        Kid& k = kids[n++];
        k.ref = nd.left;
        for (int a = 0; a < 3; ++a) { k.box.mn[a] = nd.lmin[a]; k.box.mx[a] = nd.lmax[a]; }
      }
      if (nd.right != 0xFFFFFFFFu) {
        Kid& k = kids[n++];
        k.ref = nd.right;
        for (int a = 0; a < 3; ++a) { k.box.mn[a] = nd.rmin[a]; k.box.mx[a] = nd.rmax[a]; }
      }
    };
    take(in[ref]);
    while (n < 4) {
      int open = -1;
      float largest = -1.0f;
      for (int i = 0; i < n; ++i)
        if (inner(kids[i].ref) && kids[i].box.half_area() > largest) { largest = kids[i].box.half_area(); open = i; }
      if (open < 0) break;
      // the opened child's children take its place and the next free slot (n <= 3 here, so both fit)
      const HostNode& nd = in[kids[open].ref];
      kids[open] = kids[n - 1];
      --n;
      take(nd);
    }
    const uint32_t index = (uint32_t)out.size();
    out.emplace_back();
    uint32_t refs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int i = 0; i < n; ++i) {
      uint32_t h = 0;
      refs[i] = convert(kids[i].ref, h);
      height = std::max(height, h);
    }
    WideNode& w = out[index];
    for (int i = 0; i < 4; ++i) {
      w.child[i] = refs[i];
      w.pad[i] = 0;
      for (int a = 0; a < 3; ++a) {
        w.lo[a][i] = i < n ? kids[i].box.mn[a] : std::numeric_limits<float>::infinity();
        w.hi[a][i] = i < n ? kids[i].box.mx[a] : -std::numeric_limits<float>::infinity();
      }
    }
    ++height;
    return index;
  }
};

}  // namespace

WideBvh widen_bvh(const std::vector<HostNode>& nodes, uint32_t root_ref) {
  WideBvh out;
  out.nodes.reserve(nodes.size() / 2 + 1);
  Widen w{nodes, out.nodes};
  out.root_ref = w.convert(root_ref, out.height);
  return out;
}

std::vector<QuantNode> quantize_wide(const WideBvh& wide) {
  std::vector<QuantNode> out(wide.nodes.size());
  for (size_t i = 0; i < wide.nodes.size(); ++i) {
    const WideNode& w = wide.nodes[i];
    QuantNode& q = out[i];
    std::memset(&q, 0, sizeof(q));
    for (int c = 0; c < 4; ++c) q.child[c] = w.child[c];
    for (int a = 0; a < 3; ++a) {
      float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
      for (int c = 0; c < 4; ++c)
        if (w.child[c] != 0xFFFFFFFFu) { mn = std::min(mn, w.lo[a][c]); mx = std::max(mx, w.hi[a][c]); }
      if (!(mn <= mx)) { mn = 0.0f; mx = 0.0f; }  // (no children: cannot happen for a built tree)
      q.origin[a] = mn;
      // the smallest spacing with which 255 steps reach the far side: a 255th of the extent, and then the next
      // floats up until every plane, decoded exactly as the device does, lies on the outside of its child's box
      // (a few tries; should rounding keep getting in the way, twice the spacing always fits)
      float step = std::max((mx - mn) / 255.0f, std::numeric_limits<float>::min());
      bool ok = false;
      for (int tries = 0; !ok && std::isfinite(step); ++tries) {
        ok = true;
        for (int c = 0; c < 4 && ok; ++c) {
          if (w.child[c] == 0xFFFFFFFFu) { q.lo[a][c] = 255; q.hi[a][c] = 0; continue; }
          float lo = std::floor((w.lo[a][c] - mn) / step);
          lo = std::min(std::max(lo, 0.0f), 255.0f);
          while (lo > 0.0f && std::fmaf(lo, step, mn) > w.lo[a][c]) lo -= 1.0f;
          float hi = std::ceil((w.hi[a][c] - mn) / step);
          hi = std::max(hi, 0.0f);
          while (hi <= 255.0f && std::fmaf(hi, step, mn) < w.hi[a][c]) hi += 1.0f;
          if (hi > 255.0f || std::fmaf(lo, step, mn) > w.lo[a][c]) { ok = false; break; }
          q.lo[a][c] = (uint8_t)lo;
          q.hi[a][c] = (uint8_t)hi;
        }
        if (!ok) step = tries < 16 ? std::nextafter(step, std::numeric_limits<float>::infinity()) : step * 2.0f;
      }
      if (!ok) {  // (bounds near the largest f32: no spacing fits; the whole grid then stands for every child)
        step = std::numeric_limits<float>::max() / 256.0f;
        for (int c = 0; c < 4; ++c)
          if (w.child[c] != 0xFFFFFFFFu) { q.lo[a][c] = 0; q.hi[a][c] = 255; }
      }
      (a == 0 ? q.step_x : a == 1 ? q.step_y : q.step_z) = step;
    }
  }
  return out;
}

namespace {
// (the tree is no taller than the traversal stack, so the recursion is shallow)
float growth_below(std::vector<HostNode>& nodes, uint32_t n_nodes, uint32_t n_copies, uint32_t ref,
                   const std::vector<float>& prim_growth) {
  if (ref == 0xFFFFFFFFu) return 0.0f;
  if (ref & 0x80000000u) {
    const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 7u) + 1u;
    float g = 0.0f;
    for (uint32_t k = first; k < first + count && k < prim_growth.size(); ++k) g = std::max(g, prim_growth[k]);
    return g;
  }
  const float l = growth_below(nodes, n_nodes, n_copies, nodes[ref].left, prim_growth);
  const float r = growth_below(nodes, n_nodes, n_copies, nodes[ref].right, prim_growth);
  for (uint32_t c = 0; c < n_copies; ++c) {
    nodes[(size_t)c * n_nodes + ref].lgrow = l;
    nodes[(size_t)c * n_nodes + ref].rgrow = r;
  }
  return std::max(l, r);
}
}  // namespace

void set_box_growth(std::vector<HostNode>& nodes, uint32_t n_nodes, uint32_t n_copies, uint32_t root_ref,
                    const std::vector<float>& prim_growth) {
  if (n_nodes == 0 || nodes.size() < (size_t)n_nodes * n_copies) return;
  (void)growth_below(nodes, n_nodes, n_copies, root_ref, prim_growth);
}

void refit_bvh(std::vector<HostNode>& nodes, uint32_t root_ref, const std::vector<BuildPrim>& prims) {
  Refit r{nodes, prims};
  (void)r.fit(root_ref);
}

}  // namespace mirt
