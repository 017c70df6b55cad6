// CPU check of the cuboid finder of the brute-force ray loop (drmlt-mitsuba_amd/csrc/box_merge.h). Reads parallelograms from
// stdin (one per line: corner, edge 1, edge 2 -- nine numbers; a line "x" is a record that is no parallelogram) and prints the
// cuboids found as JSON. Built and run by tests/test_box_merge.py (g++, no GPU).
#include "box_merge.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

int main(int argc, char **argv) {
    const int min_faces = argc > 1 ? atoi(argv[1]) : 4;
    std::vector<QuadGeo> quads;
    char line[1024];
    while (fgets(line, sizeof line, stdin)) {
        QuadGeo q{};
        q.usable = sscanf(line, "%lf %lf %lf %lf %lf %lf %lf %lf %lf", &q.a[0], &q.a[1], &q.a[2], &q.e1[0], &q.e1[1], &q.e1[2], &q.e2[0], &q.e2[1], &q.e2[2]) == 9;
        quads.push_back(q);
    }
    const std::vector<BoxGeo> boxes = find_boxes(quads, min_faces);
    printf("[");
    for (size_t i = 0; i < boxes.size(); ++i) {
        const BoxGeo &b = boxes[i];
        printf("%s{\"a\": [%.17g, %.17g, %.17g], \"E\": [", i ? ", " : "", b.a[0], b.a[1], b.a[2]);
        for (int k = 0; k < 3; ++k) printf("%s[%.17g, %.17g, %.17g]", k ? ", " : "", b.E[k][0], b.E[k][1], b.E[k][2]);
        printf("], \"face\": [%d, %d, %d, %d, %d, %d], \"code\": [%d, %d, %d, %d, %d, %d], \"n_faces\": %d}", b.face[0], b.face[1], b.face[2], b.face[3],
               b.face[4], b.face[5], b.code[0], b.code[1], b.code[2], b.code[3], b.code[4], b.code[5], b.n_faces);
    }
    printf("]\n");
    return 0;
}
