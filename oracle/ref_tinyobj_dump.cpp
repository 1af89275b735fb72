// oracle/ref_tinyobj_dump.cpp — mesh-fixture generator. TEST INFRASTRUCTURE ONLY.
//
// Compiles the reference's own vendored tiny_obj_loader.h where it lies
// (/root/reference/PathTracer/FireflyEngine/tiny_obj_loader.h, self-contained) and flattens an OBJ the way the
// reference's LoadMesh does (utilities.h:781-840: one Triangle per face from the face's first three indexed
// vertices, position / normal / uv looked up through the per-vertex index triple).  Unlike LoadMesh, a missing
// normal or uv index yields zeros instead of an out-of-bounds read.
//
// Output: a flat binary the tests and the bench load on the GPU box (where /root/reference does not exist):
//   int32 magic 'FFTR', int32 ntris, then ntris * 24 float32 in FfTriangle order
//   (v0 v1 v2 | uv0 uv1 uv2 | n0 n1 n2), include/firefly/ff_types.h.
//
// usage: tinyobj_dump in.obj out.fftri
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s in.obj out.fftri\n", argv[0]); return 2; }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string warn, err;
    bool ok = tinyobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, argv[1]);
    if (!ok) { fprintf(stderr, "LoadObj failed: %s\n", err.c_str()); return 1; }

    std::vector<float> out;
    int ntris = 0;
    for (const auto& shape : shapes) {
        size_t off = 0;
        for (size_t f = 0; f < shape.mesh.num_face_vertices.size(); ++f) {
            int fv = shape.mesh.num_face_vertices[f];
            float P[3][3] = {}, N[3][3] = {}, UV[3][2] = {};
            for (int v = 0; v < fv && v < 3; ++v) {
                tinyobj::index_t idx = shape.mesh.indices[off + v];
                for (int c = 0; c < 3; ++c) P[v][c] = attrib.vertices[3 * idx.vertex_index + c];
                if (idx.normal_index >= 0) for (int c = 0; c < 3; ++c) N[v][c] = attrib.normals[3 * idx.normal_index + c];
                if (idx.texcoord_index >= 0) for (int c = 0; c < 2; ++c) UV[v][c] = attrib.texcoords[2 * idx.texcoord_index + c];
            }
            off += fv;
            if (fv < 3) continue;
            for (int v = 0; v < 3; ++v) for (int c = 0; c < 3; ++c) out.push_back(P[v][c]);
            for (int v = 0; v < 3; ++v) for (int c = 0; c < 2; ++c) out.push_back(UV[v][c]);
            for (int v = 0; v < 3; ++v) for (int c = 0; c < 3; ++c) out.push_back(N[v][c]);
            ++ntris;
        }
    }
    FILE* fp = fopen(argv[2], "wb");
    if (!fp) { perror(argv[2]); return 1; }
    int32_t hdr[2] = { 0x52544646, ntris };
    fwrite(hdr, sizeof hdr, 1, fp);
    fwrite(out.data(), sizeof(float), out.size(), fp);
    fclose(fp);
    printf("%s: %d triangles -> %s\n", argv[1], ntris, argv[2]);
    return 0;
}
