/* Generates tests/golden/glibc_rand.json: the first draws of glibc's rand() for a few seeds (the generator the
 * reference's sampleMesh uses, mesh_sample.cpp:51-62). Build and run: gcc -O0 gen_glibc_rand.c -o /tmp/g && /tmp/g > glibc_rand.json */
#include <stdio.h>
#include <stdlib.h>
int main(void) {
    const unsigned seeds[] = {1u, 2u, 12345u, 4294967295u, 0u};
    printf("{\"rand_max\": %d, \"draws\": {", RAND_MAX);
    for (int s = 0; s < 5; ++s) {
        srand(seeds[s]);
        printf("%s\"%u\": [", s ? ", " : "", seeds[s]);
        for (int i = 0; i < 400; ++i) printf("%s%d", i ? ", " : "", rand());
        printf("]");
    }
    /* the unseeded stream (what the reference's process sees) */
    printf("}}\n");
    return 0;
}
