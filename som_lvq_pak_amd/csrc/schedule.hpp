// schedule.hpp -- per-iteration host scalars of the epoch loops, written with the same
// C types the reference uses so that every value is bit-identical to its own:
//   learning rate   linear_alpha / inverse_t_alpha   lvq_pak.c:903-906 / 914-921
//   radius          som_training                      som_rout.c:615
//   weights         som_training                      som_rout.c:622-624
// plus the bubble threshold that lets the device drop hexa_dist/rect_dist's sqrt.
// Compile with -ffp-contract=off (the build does).
#pragma once
#include <cmath>
#include <cstdint>

namespace somhip {

inline float alpha_at(int type, int64_t iter, int64_t length, float alpha) {
  if (type == 2) {                                   // inverse_t, constant 100.0 (lvq_pak.c:909)
    float c = static_cast<float>(length) / 100.0f;
    float num = alpha * c;
    return num / (c + static_cast<float>(iter));
  }
  float num = alpha * static_cast<float>(length - iter);
  return num / static_cast<float>(length);
}

inline float radius_at(int64_t iter, int64_t length, float radius) {
  double r = (static_cast<double>(radius) - 1.0) * static_cast<double>(static_cast<float>(length - iter));
  r = r / static_cast<double>(static_cast<float>(length));
  return static_cast<float>(1.0 + r);
}

inline float weighted_alpha(float talp, float weight) {
  float p = static_cast<float>(std::pow(1.0 - static_cast<double>(talp), static_cast<double>(weight)));
  return static_cast<float>(1.0 - static_cast<double>(p));
}

// The reference tests   (float)sqrt((double)ret) <= radius   (som_rout.c:452,496) where
// `ret` is the squared lattice distance the device forms exactly (kernels.hpp
// lattice_sq).  r -> (float)sqrt((double)r) is monotone, so the set of passing `ret`
// values is { ret <= T }; T is found here with the host's correctly rounded sqrt and
// the device only compares.  radius < 0 admits nothing (sqrt >= 0).
inline float bubble_threshold(float radius) {
  if (!(radius >= 0.0f)) return -1.0f;
  auto inside = [radius](float r) { return static_cast<float>(std::sqrt(static_cast<double>(r))) <= radius; };
  float t = static_cast<float>(static_cast<double>(radius) * static_cast<double>(radius));
  if (std::isinf(t)) return t;
  while (t > 0.0f && !inside(t)) t = std::nextafterf(t, -INFINITY);
  for (;;) {
    float up = std::nextafterf(t, INFINITY);
    if (std::isinf(up) || !inside(up)) break;
    t = up;
  }
  return t;
}

}  // namespace somhip
