// Generates tests/golden/kat1_inputs.json: the observation map the reference's kernel test draws
// (test/test_obs_to_mod_kernels.cpp:42-53): a default-constructed std::default_random_engine, bernoulli_distribution(0.8)
// for validity and normal_distribution<float>(0, 0.1) for the coordinates of a 10x10 vertex map. Only the C++ standard
// library runs here (libstdc++ of this image, the library the reference's host compiler links); no reference code.
//
// The reference writes  make_float4(normal(gen), normal(gen), normal(gen), 1) : the order in which the three draws land in
// x, y, z is the compiler's argument evaluation order (unspecified by the language). "as_compiled" is what g++ does with
// that very expression (the reference's .cpp tests are built by the host g++, CMakeLists.txt:303-341); "left_to_right" is
// the other possibility, kept so that the tests cover both.
//   g++ -O1 -o /tmp/gen_kat1 tests/golden/gen_kat1_inputs.cpp && /tmp/gen_kat1 > tests/golden/kat1_inputs.json
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

struct float4_ { float x, y, z, w; };
static inline float4_ make_float4_(float x, float y, float z, float w) { float4_ r = {x, y, z, w}; return r; }
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void emit(const char *name, bool asCompiled, int n, bool last) {
    std::default_random_engine generator;
    std::bernoulli_distribution validDistribution(0.8);
    std::normal_distribution<float> normalDistribution(0, 0.1);
    printf("  \"%s\": [", name);
    for (int i = 0; i < n; ++i) {
        float4_ v;
        if (validDistribution(generator)) {
            if (asCompiled) {
                v = make_float4_(normalDistribution(generator), normalDistribution(generator), normalDistribution(generator), 1);
            } else {
                const float a = normalDistribution(generator);
                const float b = normalDistribution(generator);
                const float c = normalDistribution(generator);
                v = make_float4_(a, b, c, 1);
            }
        } else {
            v = make_float4_(0, 0, 0, 0);
        }
        printf("%s[%u, %u, %u, %u]", i ? ", " : "", bits(v.x), bits(v.y), bits(v.z), bits(v.w));
    }
    printf("]%s\n", last ? "" : ",");
}

int main() {
    printf("{\n  \"width\": 10, \"height\": 10, \"encoding\": \"uint32 bit patterns of float4 (x, y, z, w) per pixel, row-major\",\n");
    printf("  \"compiler\": \"g++ %s\",\n", __VERSION__);
    emit("as_compiled", true, 100, false);
    emit("left_to_right", false, 100, true);
    printf("}\n");
    return 0;
}
