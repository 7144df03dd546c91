/* visfs_window.h — C ABI of the sliding-window container that feeds the bundle-adjustment backend
 * (SURVEY §8f rows f1 "LocalMap → flat-graph packer / un-packer" and f2 "wheel-odometry link generation").
 *
 * Replaces, for the BA path only, VISFS::Map::LocalMap (corelib/include/LocalMap.h:56-113, corelib/src/LocalMap.cpp):
 *   visfs_window_insert     LocalMap::insertSignature                 LocalMap.cpp:48-131
 *   visfs_window_remove     LocalMap::removeSignature                 LocalMap.cpp:133-168
 *   visfs_window_available  LocalMap::checkMapAvaliable               LocalMap.cpp:296-302
 *   visfs_window_build      getSignaturePoses + getSignatureLinks + getFeaturePosesAndObservations
 *                                                                     LocalMap.cpp:228-294 (consumer Estimator.cpp:227-254)
 *   visfs_window_update     LocalMap::updateLocalMap                  LocalMap.cpp:170-226 (caller Estimator.cpp:275-317)
 * Host-only code (no GPU): the window it emits is the `visfs_ba_window` of visfs_ba.h, consumed by
 * visfs_ba_solve_window.  Plain pointers and sizes; no exceptions cross the boundary.
 */
#ifndef VISFS_WINDOW_H
#define VISFS_WINDOW_H

#include <stdint.h>
#include "visfs_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VISFS_WINDOW_ABI_VERSION 1

typedef struct visfs_window_map visfs_window_map;

/* keys/values: the reference's ParametersMap entries (LocalMap/MapSize, Tracker/MaxFeatures, LocalMap/MinParallax,
 * LocalMap/MinTranslation, Estimator/MinInliers); absent keys take the reference defaults (Parameters.h:148,161-163,171). */
int  visfs_window_abi_version(void);
int  visfs_window_create(int n_params, const char* const* keys, const char* const* values, visfs_window_map** out);
void visfs_window_destroy(visfs_window_map* m);

/* One new signature.  word_*: Signature::getWords / getKeyPointsMatchesImageRight / getWords3d, ascending word id;
 * word_uv = [n][4] (u, v, u_right, v_right) floats, word_xyz = [n][3] robot-frame floats (may be non-finite),
 * word_has3d = id present in words3d.  cov_*: Signature::getCovisibleWords (key-points in the former signature).
 * wheel_odom = all zeros when wheel odometry is unavailable (the reference's zero-matrix sentinel, LocalMap.cpp:256).
 * Returns 1 if inserted, 0 if refused (no 3-D words, LocalMap.cpp:49-52), <0 on bad arguments. */
int  visfs_window_insert(visfs_window_map* m, uint64_t signature_id, const double pose_Twr[12], const double wheel_odom[12],
                         const double translation[3],
                         int32_t n_words, const uint64_t* word_ids, const float* word_uv, const float* word_xyz, const uint8_t* word_has3d,
                         int32_t n_cov, const uint64_t* cov_ids, const float* cov_uv);
void visfs_window_remove(visfs_window_map* m);
int  visfs_window_available(const visfs_window_map* m);
int  visfs_window_is_key_signature(const visfs_window_map* m);

/* Fills *out with pointers into the map's own buffers (valid until the next mutating call).  with_links mirrors
 * `sensorStrategy_ >= 2` (Estimator.cpp:235-236); root id = newest signature id - 1 (Estimator.cpp:252). */
int  visfs_window_build(visfs_window_map* m, const double Trc[12], double fx, double fy, double cx, double cy, float baseline,
                        int32_t n_cameras, int32_t with_links, visfs_ba_window* out);

/* Results back into the window.  error_vertex receives the ids of features to block (c1 && c2 && c3, LocalMap.cpp:207-221),
 * ascending, at most `capacity`; *n_error = number found. */
int  visfs_window_update(visfs_window_map* m, int32_t n_poses, const uint64_t* pose_ids, const double* pose_Twr,
                         int32_t n_points, const uint64_t* point_ids, const double* point_xyz,
                         int32_t n_outliers, const uint64_t* outlier_feature, const uint64_t* outlier_pose,
                         uint64_t* error_vertex, int32_t capacity, int32_t* n_error);
/* Same, taking the visfs_ba_result of the window built last (points are read from that window's in/out array). */
int  visfs_window_apply(visfs_window_map* m, const visfs_ba_result* result, uint64_t* error_vertex, int32_t capacity, int32_t* n_error);

/* ---- introspection (tests) */
int  visfs_window_counts(const visfs_window_map* m, int32_t* n_signatures, int32_t* n_features, int32_t* n_observations);
int  visfs_window_counters(const visfs_window_map* m, int32_t* new_features, int32_t* signatures, float* parallax, double translation[3]);
/* Arrays sized from visfs_window_counts; obs_vals = [n_observations][7] (u v u_right v_right x y z). */
int  visfs_window_dump(const visfs_window_map* m, uint64_t* sig_ids, double* sig_pose,
                       uint64_t* feat_ids, uint64_t* feat_start, uint64_t* feat_end, int32_t* feat_state, double* feat_xyz,
                       int32_t* feat_nobs, uint64_t* obs_sig, float* obs_vals);

#ifdef __cplusplus
}
#endif
#endif
