/* hrt_io.h -- host-side readers for the data formats on the input side of the path (C ABI, no GPU needed).
 *
 * They replace, for a headless host, what the reference reads through VTK and nlohmann::json:
 *   hrt_io_read_stl            vtk_reader::readSTLFile, src/Util/VTKReaderImpl.cpp:254-318 (ASCII STL)
 *   hrt_io_read_particle_vtk   vtk_reader::readVTKTimeFile, src/Util/VTKReaderImpl.cpp:139-252 (legacy ASCII POLYDATA)
 *   hrt_io_read_series         VTKTimeReader::readSeriesFile, src/Util/VTKTimeReader.cu:31-88
 *   hrt_io_bake_color_ramp     bakeColorRamp / colorStopsForPreset, include/Util/ColorRamp.cuh:31-112
 *   hrt_io_construct_transform MathHelper::constructTransformMatrix, include/Global/DeviceFunctions.cuh:133-148 (host, float libm)
 *   hrt_io_load_config         ProgramArgumentParser::parseProgramArguments, src/Util/ProgramArgumentParser.cu:4-165
 *   hrt_io_read_mesh_cache / hrt_io_write_mesh_cache   Mesh-mode particleN.cache, src/Util/VTKMeshReader.cu:54-72,217-257
 *   hrt_io_read_metadata_cache / hrt_io_write_metadata_cache   metadata.cache, src/Util/VTKMeshReader.cu:196-205,273-282
 *   hrt_io_read_vtk_mesh_file   Mesh-mode VTK files (triangle strips + cell data), src/Util/VTKReaderImpl.cpp:24-137
 * Every function returns 0 on success; hrt_io_last_error() describes the last failure of the calling thread
 * (the reference logs and exit()s instead: src/Util/VTKReaderImpl.cpp, VTK_READER_ERROR_EXIT_CODE).
 */
#ifndef HRT_IO_H
#define HRT_IO_H

#include <stdint.h>
#include "hrt_params.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *hrt_io_last_error(void);

/* One STL shape.  vertices: 9 floats per triangle in file order (float(double) as the reference converts them).
 * normals: 9 floats per triangle -- the unit geometric normal of the facet replicated for its three vertices, i.e. the
 * layout the shader indexes (3*prim + k, shader/Shader.cu:140-142), oriented as vtkPolyDataNormals orients it with the
 * reference's settings (Consistency + AutoOrientNormals, src/Util/VTKReaderImpl.cpp:279-285: coincident points merged,
 * every connected component made consistent across its manifold edges and turned outwards from its leftmost triangle);
 * the vertices keep the file's order, as in the reference.  The reference feeds one normal per FACE here (quirk Q5: wrong
 * normals and out-of-bounds reads); this is the per-vertex array Mesh mode uses.
 * file_normals: 3 floats per triangle, the "facet normal" lines (vtkSTLReader ignores them; kept for checks). */
typedef struct HrtIoMesh {
    float   *vertices, *normals, *file_normals;
    uint64_t n_triangles;
} HrtIoMesh;
int  hrt_io_read_stl(const char *path, HrtIoMesh *out);
void hrt_io_free_mesh(HrtIoMesh *mesh);

/* One particle VTK file: states[i] = {quat (file components 0..3 in x,y,z,w), position, velocity}. */
typedef struct HrtIoParticles {
    HrtParticleState *states;
    uint64_t *ids, *shape_ids;
    uint64_t  n;
} HrtIoParticles;
int  hrt_io_read_particle_vtk(const char *path, HrtIoParticles *out);
void hrt_io_free_particles(HrtIoParticles *p);

/* *.vtk.series: file paths (directory + name) and per-file durations: t[i+1] - t[i], the last one repeats the
 * one before it, a single entry lasts 1000 (VTKTimeReader.cu:72-82). */
typedef struct HrtIoSeries {
    char   **files;
    float   *durations;
    uint64_t n;
} HrtIoSeries;
int  hrt_io_read_series(const char *directory, const char *name, HrtIoSeries *out);
void hrt_io_free_series(HrtIoSeries *s);

/* count RGB colours of a preset ("viridis" "plasma" "spectral" "terrain" "heatmap" "grayscale", case-insensitive,
 * anything else = viridis as resolveColorRampPreset does). */
int  hrt_io_bake_color_ramp(const char *preset, uint64_t count, float *out_rgb);

int  hrt_io_construct_transform(const float *shift3, const float *rotate_deg3, const float *scale3, float *out12);

typedef struct HrtIoSphere {
    float    center[3], radius;
    int32_t  metal;                 /* mat-type: 0 ROUGH, 1 METAL */
    uint64_t material_index;
    float    transform[12];         /* constructTransformMatrix(shift, rotate, scale) */
} HrtIoSphere;

typedef struct HrtIoConfig {
    int32_t  mesh, cache, debug_mode, api_is_opengl;
    char    *series_path, *series_name, *cache_path, *stl_path, *particle_material_preset, *api;
    uint64_t cache_process_thread_count;
    float   *roughs;  uint64_t n_roughs;           /* 3 floats each */
    float   *metals;  uint64_t n_metals;           /* 4 floats each: albedo, fuzz */
    HrtIoSphere *spheres; uint64_t n_spheres;
    int32_t  window_width, window_height;
    uint64_t fps, render_speed_ratio, camera_initial_speed_ratio;
    float    camera_center[3], camera_target[3], up_direction[3];
    float    particle_shift[3], particle_scale[3];
    float    mouse_sensitivity, camera_pitch_limit_degree, camera_speed_stride;
} HrtIoConfig;
int  hrt_io_load_config(const char *path, HrtIoConfig *out);
void hrt_io_free_config(HrtIoConfig *c);

/* Mesh-mode cache file of one VTK time step (VTKMeshReader.cuh:15-23): all particles of the file with their
 * triangles.  Arrays are concatenated over particles; particle p owns triangles [first[p], first[p+1]). */
typedef struct HrtIoMeshCache {
    uint64_t  n_particles;
    uint64_t *ids;
    float    *velocities;           /* 3 per particle */
    uint64_t *first_triangle;       /* n_particles + 1 */
    float    *vertices, *normals;   /* 9 per triangle each */
} HrtIoMeshCache;
int  hrt_io_read_mesh_cache(const char *path, HrtIoMeshCache *out);
int  hrt_io_write_mesh_cache(const char *path, const HrtIoMeshCache *in);
void hrt_io_free_mesh_cache(HrtIoMeshCache *c);

/* metadata.cache of a cache directory (VTKMeshReader.cuh:23; written at VTKMeshReader.cu:196-205, read at :273-282): the
 * largest cell count of any VTK file of the series, as decimal text -- it sizes the material array of Mesh mode.
 * `directory` ends with a separator, as the reference's cache path does. */
int  hrt_io_read_metadata_cache(const char *directory, uint64_t *out_max_cell_count);
int  hrt_io_write_metadata_cache(const char *directory, uint64_t max_cell_count);

/* Mesh-mode VTK file with embedded geometry (vtk_reader::readVTKMeshFile, src/Util/VTKReaderImpl.cpp:24-137): legacy ASCII
 * POLYDATA whose cells are TRIANGLE_STRIPS, one strip per particle, with CELL_DATA `id` and `vel`.  Strip k yields
 * (points - 2) triangles, odd ones with their last two vertices swapped (:96-104); every triangle vertex carries its POINT's
 * normal = the normalised sum of the unit normals of the triangles that use the point (vtkPolyDataNormals with point normals,
 * no splitting), the triangles first made consistent and oriented outwards per connected component as that filter does with
 * the reference's settings (:54-60).  FIELD and METADATA blocks are read past, as vtkPolyDataReader does.  The result
 * has the layout of a cache file, so hrt_io_write_mesh_cache(out) writes what the reference's cache run writes.
 * *out_cell_count (may be NULL) receives the file's cell count (the reference's maxCellCountSingleFile candidate). */
int  hrt_io_read_vtk_mesh_file(const char *path, HrtIoMeshCache *out, uint64_t *out_cell_count);

#ifdef __cplusplus
}
#endif
#endif /* HRT_IO_H */
