/*
 * FastClusterWrapper.h — link-compatible declaration of the one FFI symbol the Swift side of
 * FluidAudio binds for clustering.  Putting this header (and module.modulemap next to it) in
 * place of the reference's Sources/FastClusterWrapper/include/ lets
 * Diarizer/Offline/Clustering/AHCClustering.swift:40-50 compile and link unchanged against
 * libfluidaudio_hip.so.
 *
 * Replaces: Sources/FastClusterWrapper/include/FastClusterWrapper.h:11-19 (status enum),
 *           :35-41 (entry point); behaviour contract: Sources/FastClusterWrapper/FastClusterWrapper.cpp:196-244.
 */
#ifndef FASTCLUSTER_WRAPPER_H
#define FASTCLUSTER_WRAPPER_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    FASTCLUSTER_WRAPPER_SUCCESS = 0,            /* dendrogram fully written (or N <= 1: nothing to write) */
    FASTCLUSTER_WRAPPER_INVALID_ARGUMENT = 1,   /* NULL pointer, or dimension == 0 with pointCount > 0 */
    FASTCLUSTER_WRAPPER_INDEX_OVERFLOW = 2,     /* pointCount or dimension > INT32_MAX */
    FASTCLUSTER_WRAPPER_OUTPUT_TOO_SMALL = 3,   /* dendrogramLength < (pointCount-1)*4 */
    FASTCLUSTER_WRAPPER_ALLOCATION_FAILURE = 4, /* host or HBM allocation failed */
    FASTCLUSTER_WRAPPER_RUNTIME_ERROR = 5,      /* NaN in a distance, or a HIP runtime error */
    FASTCLUSTER_WRAPPER_UNKNOWN_ERROR = 255
} fastcluster_wrapper_status;

/* Centroid-linkage (UPGMC) dendrogram of pointCount row-major fp64 vectors.
 * dendrogramOut receives (pointCount-1) rows of (left, right, distance, size) in MERGE order,
 * left < right, new node ids N, N+1, ...  Both buffers are HOST memory owned by the caller;
 * the call is synchronous and re-entrant. */
fastcluster_wrapper_status fastcluster_compute_centroid_linkage(
    const double *data, size_t pointCount, size_t dimension, double *dendrogramOut, size_t dendrogramLength);

#ifdef __cplusplus
}
#endif
#endif
