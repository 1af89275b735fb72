// oracle/ref_glm_vectors.cpp — golden-vector generator. TEST INFRASTRUCTURE ONLY.
//
// Compiles against the reference's OWN vendored glm 0.9.9.7 headers where they lie
// (/root/reference/external/include/glm-0.9.9.7, header-only, no stand-ins needed) and dumps the results of
// every glm routine the hot path depends on (SURVEY.md §8a "glm semantics") for seeded random inputs.
// The output (tests/golden/glm_vectors.bin) pins the oracle's and the library's glm restatements bit-for-bit.
//
// Built and run by `make -C oracle golden` in the build container only; the binary lands in oracle/_ref/
// (git-ignored).  Nothing here travels to the GPU box except the generated fixture.
//
// Record layout (float32, little-endian), NCASE records of NIN + NOUT floats after a 4-int header
// {magic 'GLMV', NCASE, NIN, NOUT}:
//   in : A[16] B[16] v[4] a[3] b[3] angle persp[5]={fov_deg,w,h,near,far}                       (48)
//   out: A*v[4] A*B[16] inverse(A)[16] transpose(A)[16] translate(A,a)[16] rotate(A,angle,b)[16]
//        scale(A,a)[16] lookAtRH(a,b,v.xyz)[16] perspectiveFovRH(radians(fov),w,h,n,f)[16]
//        normalize(a)[3] cross(a,b)[3] dot(a,b) distance(a,b) radians(angle)
//        (inverse(transpose(A))*vec4(a,0))[4] normalize(A*vec4(b,0))[4]
//        model=T(a)*Rx*Ry*Rz(90*b)*S(|v.xyz|+0.5)[16] inverse(model)[16]
//        inverse(lookAtRH(a,a+normalize(b),up)) * inverse(perspectiveFovRH(...))[16]                (197)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "glm.hpp"
#include "gtc/matrix_transform.hpp"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd_u32()
{
    // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (uint32_t)((z ^ (z >> 31)) >> 32);
}
static float rnd(float lo, float hi) { return lo + (hi - lo) * (float)(rnd_u32() >> 8) * (1.0f / 16777216.0f); }

static void put(std::vector<float>& o, const glm::mat4& m) { for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) o.push_back(m[c][r]); }
static void put(std::vector<float>& o, const glm::vec4& v) { for (int i = 0; i < 4; ++i) o.push_back(v[i]); }
static void put(std::vector<float>& o, const glm::vec3& v) { for (int i = 0; i < 3; ++i) o.push_back(v[i]); }

int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "glm_vectors.bin";
    const int NCASE = 256, NIN = 48, NOUT = 197;
    std::vector<float> out;
    for (int k = 0; k < NCASE; ++k) {
        glm::vec3 a(rnd(-5, 5), rnd(-5, 5), rnd(-5, 5));
        glm::vec3 b(rnd(-2, 2), rnd(-2, 2), rnd(-2, 2));
        glm::vec4 v(rnd(-3, 3), rnd(-3, 3), rnd(-3, 3), rnd(-3, 3));
        float angle = rnd(-180, 180);
        float persp[5] = { rnd(20, 110), (float)(16 * (1 + rnd_u32() % 240)), (float)(16 * (1 + rnd_u32() % 135)), rnd(0.01f, 1.f), rnd(50.f, 2000.f) };
        glm::mat4 A, B;
        if (k & 1) {
            // affine TRS-like matrices
            A = glm::translate(glm::mat4(1.0f), a) * glm::rotate(glm::mat4(1.0f), glm::radians(angle), glm::normalize(b + glm::vec3(0.1f, 0.2f, 0.3f))) * glm::scale(glm::mat4(1.0f), glm::vec3(rnd(0.2f, 6), rnd(0.2f, 6), rnd(0.2f, 6)));
            B = glm::translate(glm::mat4(1.0f), b) * glm::scale(glm::mat4(1.0f), glm::vec3(rnd(0.5f, 2), rnd(0.5f, 2), rnd(0.5f, 2)));
        } else {
            for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) { A[c][r] = rnd(-2, 2) + (c == r ? 3.0f : 0.0f); B[c][r] = rnd(-2, 2); }
        }
        size_t start = out.size();
        put(out, A); put(out, B); put(out, v); put(out, a); put(out, b); out.push_back(angle);
        for (float p : persp) out.push_back(p);
        if ((int)(out.size() - start) != NIN) { fprintf(stderr, "NIN mismatch %zu\n", out.size() - start); return 2; }

        put(out, A * v);
        put(out, A * B);
        put(out, glm::inverse(A));
        put(out, glm::transpose(A));
        put(out, glm::translate(A, a));
        put(out, glm::rotate(A, angle, b));
        put(out, glm::scale(A, a));
        put(out, glm::lookAtRH(a, b, glm::vec3(v)));
        put(out, glm::perspectiveFovRH(glm::radians(persp[0]), persp[1], persp[2], persp[3], persp[4]));
        put(out, glm::normalize(a));
        put(out, glm::cross(a, b));
        out.push_back(glm::dot(a, b));
        out.push_back(glm::distance(a, b));
        out.push_back(glm::radians(angle));
        put(out, glm::inverse(glm::transpose(A)) * glm::vec4(a, 0.f));
        put(out, glm::normalize(A * glm::vec4(b, 0.f)));
        {
            // the Geometry constructor's call chain (utilities.h:180-189) expressed with glm
            glm::vec3 rot = 90.f * b, scl = glm::abs(glm::vec3(v)) + 0.5f;
            glm::mat4 T = glm::translate(glm::mat4(1.0f), a);
            glm::mat4 R = glm::rotate(glm::mat4(1.0f), glm::radians(rot.x), glm::vec3(1.f, 0.f, 0.f));
            R *= glm::rotate(glm::mat4(1.0f), glm::radians(rot.y), glm::vec3(0.f, 1.f, 0.f));
            R *= glm::rotate(glm::mat4(1.0f), glm::radians(rot.z), glm::vec3(0.f, 0.f, 1.f));
            glm::mat4 S = glm::scale(glm::mat4(1.0f), scl);
            glm::mat4 model = T * R * S;
            put(out, model);
            put(out, glm::inverse(model));
        }
        {
            // the camera matrix product of kernel.cu:203 / utilities.h:299-317 expressed with glm
            glm::vec3 fwd = glm::normalize(b);
            glm::vec3 right = glm::normalize(glm::cross(fwd, glm::vec3(0.f, 1.f, 0.f)));
            glm::vec3 up = glm::normalize(glm::cross(right, fwd));
            glm::mat4 iv = glm::inverse(glm::lookAtRH(a, a + fwd, up));
            glm::mat4 ip = glm::inverse(glm::perspectiveFovRH(glm::radians(persp[0]), persp[1], persp[2], persp[3], persp[4]));
            put(out, iv * ip);
        }
        if ((int)(out.size() - start) != NIN + NOUT) { fprintf(stderr, "NOUT mismatch %zu\n", out.size() - start - NIN); return 2; }
    }
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    int32_t hdr[4] = { 0x564D4C47, NCASE, NIN, NOUT };
    fwrite(hdr, sizeof hdr, 1, f);
    fwrite(out.data(), sizeof(float), out.size(), f);
    fclose(f);
    printf("wrote %s: %d cases x (%d in + %d out) floats\n", path, NCASE, NIN, NOUT);
    return 0;
}
