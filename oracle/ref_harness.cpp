/*
 * ref_harness.cpp -- C entry points around the REAL reference translation units that build in this
 * image without any third-party library:
 *     /root/reference/src/obvision/registration/icp/assign/PairAssignment.cpp
 *     /root/reference/src/obvision/registration/icp/assign/filter/DistanceFilter.cpp
 *     /root/reference/src/obvision/registration/icp/assign/filter/ReciprocalFilter.cpp
 *     /root/reference/src/obcore/math/mathbase.h            (header-only templates)
 *     /root/reference/src/obcore/base/tools.cpp             (getDoubleLine / getIntLine of the grid text format)
 * They are compiled where they lie (oracle/Makefile target `_ref`), never copied; the output
 * oracle/_ref/libtsd_ref.so is git-ignored.  Everything else on the hot path needs GSL / FLANN
 * (absent here, and stand-ins are not allowed), so only the ICP post-assignment chain
 * and the mathbase helpers are pinned through this library.
 *
 * TEST INFRASTRUCTURE ONLY (see tsd_oracle.h).  This file is original code: it only *uses* the
 * reference's public extension point obvious::PairAssignment (the same one FlannPairAssignment,
 * AnnPairAssignment, ... derive from) with an exact brute-force nearest-neighbour search.
 */
#include "obvision/registration/icp/assign/PairAssignment.h"
#include "obvision/registration/icp/assign/filter/DistanceFilter.h"
#include "obvision/registration/icp/assign/filter/ReciprocalFilter.h"
#include "obcore/math/mathbase.h"
#include "obcore/base/tools.h"

#include <sstream>

#include <cmath>
#include <cstring>
#include <limits>

namespace {

/* exact 1-NN, first minimum, squared L2 accumulated x then y -- what FlannPairAssignment's
 * kd-tree (eps = 0) returns up to tie-breaking (FlannPairAssignment.cpp:64-92) */
class BruteForceAssignment : public obvious::PairAssignment
{
public:
  BruteForceAssignment() : obvious::PairAssignment(2) {}
  void setModel(double** model, int size) override { _model = model; _sizeModel = size; }
  void determinePairs(double** scene, bool* mask, int size) override
  {
    for(int i = 0; i < size; i++)
    {
      if(mask[i] == 1)
      {
        int best = -1;
        double bd = std::numeric_limits<double>::infinity();
        for(int k = 0; k < _sizeModel; k++)
        {
          const double dx = scene[i][0] - _model[k][0];
          const double dy = scene[i][1] - _model[k][1];
          const double d  = dx * dx + dy * dy;
          if(d < bd) { bd = d; best = k; }
        }
        addPair(best, i, bd);
      }
      else
        addNonPair(i);
    }
  }
  using obvious::PairAssignment::determinePairs;
};

/* stands in for OutOfBoundsFilter2D (needs obvious::Matrix -> GSL): the caller supplies the mask */
class ExternalMask : public obvious::IPreAssignmentFilter
{
public:
  const unsigned char* ext = nullptr;
  void filter(double**, unsigned int size, bool* mask) override
  {
    if(!ext) return;
    for(unsigned int i = 0; i < size; i++)
      if(!ext[i]) mask[i] = false;
  }
};

struct Chain
{
  BruteForceAssignment assigner;
  ExternalMask pre;
  obvious::DistanceFilter* dist;
  obvious::ReciprocalFilter recip;
};

} // namespace

extern "C" {

/* mirrors ThreadLocalize.cpp:211-220: DistanceFilter(distFilterMax, distFilterMin, icpIterations - 10)
 * with icpIterations an int (the int -> unsigned conversion happens at this call, as there) */
void* ref_chain_create(double dist_max, double dist_min, int icp_iterations)
{
  Chain* c = new Chain();
  c->dist = new obvious::DistanceFilter(dist_max, dist_min, icp_iterations - 10);
  c->assigner.addPreFilter(&c->pre);
  c->assigner.addPostFilter(c->dist);
  c->assigner.addPostFilter(&c->recip);
  return c;
}

void ref_chain_destroy(void* h)
{
  Chain* c = static_cast<Chain*>(h);
  delete c->dist;
  delete c;
}

/* Icp::reset -> PairAssignment::reset -> DistanceFilter::reset */
void ref_chain_reset(void* h) { static_cast<Chain*>(h)->assigner.reset(); }

/* one PairAssignment::determinePairs(scene, size) call; returns the pairs handed to the estimator */
int ref_chain_pairs(void* h, const double* model_xy, int n_model, const double* scene_xy, int n_scene,
                    const unsigned char* premask, int* pair_model, int* pair_scene)
{
  Chain* c = static_cast<Chain*>(h);
  double** m = new double*[n_model > 0 ? n_model : 1];
  double** s = new double*[n_scene > 0 ? n_scene : 1];
  for(int i = 0; i < n_model; i++) m[i] = const_cast<double*>(model_xy + 2 * i);
  for(int i = 0; i < n_scene; i++) s[i] = const_cast<double*>(scene_xy + 2 * i);
  c->assigner.setModel(m, n_model);
  c->pre.ext = premask;
  c->assigner.determinePairs(s, n_scene);
  std::vector<obvious::StrCartesianIndexPair>* p = c->assigner.getPairs();
  const int n = static_cast<int>(p->size());
  for(int i = 0; i < n; i++)
  {
    pair_model[i] = static_cast<int>((*p)[i].indexFirst);
    pair_scene[i] = static_cast<int>((*p)[i].indexSecond);
  }
  delete[] m;
  delete[] s;
  return n;
}

/* mathbase.h helpers used on the path */
void   ref_minmax4(const int* a, int* mn, int* mx) { obvious::minmaxArray<int>(a, 4, mn, mx); }
double ref_euklid2(const double* a, const double* b)
{
  double x[2] = {a[0], a[1]}, y[2] = {b[0], b[1]};
  return obvious::euklideanDistance<double>(x, y, 2);
}
double ref_dist_sqr2d(const double* a, const double* b) { return obvious::distSqr2D<double>(a, b); }
void   ref_norm2(double* n) { obvious::norm2<double>(n); }
double ref_deg2rad(double d) { return obvious::deg2rad(d); }

} // extern "C"

/* obvious::getDoubleLine / getIntLine (obcore/base/tools.cpp:190-215), the line readers of TsdGrid's file constructor
 * (TsdGrid.cpp:25-110): `text` is read line by line, line i as a double (kinds[i] == 0) or as an int (1). */
extern "C" void ref_text_lines(const char* text, const int* kinds, int n, double* out)
{
  std::istringstream in(text);
  for(int i = 0; i < n; i++)
    out[i] = kinds[i] ? (double)obvious::getIntLine(in) : obvious::getDoubleLine(in);
}
