/* Host-side set-up helpers of the MI355X TDGL path: Delaunay triangulation and the Voronoi dual mesh.
 *
 * These run on the CPU before the first time step (`libtdgl_mesh.so`, plain C++, no HIP): they
 * replace the two largest items of the set-up at a million sites, SciPy's Qhull call (7.7 s) and the
 * NumPy construction of the dual mesh (2.2 s).  The reference meshes with meshpy/Triangle
 * (tdgl/device/meshing.py:15-123) and builds the dual mesh in tdgl/finite_volume/mesh.py:104-151 and
 * tdgl/finite_volume/util.py:20-277.
 *
 * All functions return 0 on success and a negative code otherwise; nothing is allocated on behalf of
 * the caller.
 */
#ifndef TDGL_HOST_MESH_H
#define TDGL_HOST_MESH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDGL_MESH_OK 0
#define TDGL_MESH_ERR_ARG (-1)        /* null pointer, fewer than three points, a non-finite coordinate */
#define TDGL_MESH_ERR_DEGENERATE (-2) /* all points collinear (or coincident) */
#define TDGL_MESH_ERR_SKIPPED (-3)    /* points that coincide with another point were left out; the triangulation of the
                                         distinct points is returned */
#define TDGL_MESH_ERR_INDEX (-4)      /* a triangle refers to a site that does not exist */
#define TDGL_MESH_ERR_RESOURCES (-5)  /* out of memory inside the call (no C++ exception leaves the library) */

/* Delaunay triangulation of `n` points `xy[2 i], xy[2 i + 1]` (sweep-hull insertion in order of distance from
 * a seed circumcentre, edge flips with exact orientation / in-circle predicates: a floating-point filter first,
 * exact expansion arithmetic when the filter cannot decide).  `triangles` must hold 3 * (2 n - 5) entries;
 * triangle t is (triangles[3 t], [3 t + 1], [3 t + 2]), counter-clockwise, `*n_triangles` of them.  Four or more
 * cocircular points are triangulated one of the valid ways.  The result covers the convex hull (like
 * scipy.spatial.Delaunay(points).simplices, which it replaces in tdgl_amd.meshgen.triangulate). */
int tdgl_host_delaunay(int64_t n, const double *xy, int64_t *triangles, int64_t *n_triangles);

/* 1 if no site lies strictly inside the circumcircle of the triangle across any interior edge (the local
 * Delaunay condition, which implies the global one for a triangulation of a convex region), 0 if one does,
 * negative on bad arguments.  Exact predicates.  Used by the tests. */
int tdgl_host_is_delaunay(int64_t n, const double *xy, int64_t n_triangles, const int64_t *triangles);

/* The Voronoi dual of a triangulation, as tdgl/finite_volume/mesh.py:104-151 builds it:
 *   edges[2 m]            sorted unique site pairs (lower index first; ascending by (lower, upper));
 *   is_boundary[m]        1 for an edge with a single triangle;
 *   tri_edge[3 t]         edge index of the local edges (0,1), (1,2), (2,0) of every triangle;
 *   centers[2 m], directions[2 m], edge_lengths[m]   midpoint, (upper - lower) site vector and its length;
 *   circumcenters[2 t]    Voronoi vertices;
 *   dual_lengths[m]       |cc_a - cc_b| of the two triangles on an interior edge, |cc_a - midpoint| on a boundary edge;
 *   areas[n]              Voronoi cell areas: signed kites (site, edge midpoint, circumcentre), summed in the order
 *                         local edge 0 of all triangles, then 1, then 2 -- the order of the NumPy construction this
 *                         replaces, so the last bit of every area is the same;
 *   suspicious[n]         1 for sites with a circumcentre on the far side of an incident edge (their areas are
 *                         rebuilt by the caller the way the reference does, tdgl/finite_volume/util.py:169-277).
 * `*n_edges` receives m; the per-edge arrays must hold 3 t edges (a triangulation never has more). */
int tdgl_host_dual_mesh(int64_t n, const double *xy, int64_t n_triangles, const int64_t *triangles, int64_t *n_edges,
                        int64_t *edges, uint8_t *is_boundary, int64_t *tri_edge, double *centers, double *directions,
                        double *edge_lengths, double *circumcenters, double *dual_lengths, double *areas,
                        uint8_t *suspicious);

#ifdef __cplusplus
}
#endif
#endif
