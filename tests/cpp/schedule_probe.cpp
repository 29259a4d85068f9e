// prints IisptSchedule's tasks (csrc/host/gpu_iispt_integrator.h) for tests/test_iispt_host.py: schedule_probe x0 y0 x1 y1 n
#include <cstdio>
#include <cstdlib>

#include "../../pbrt-v3-iile_amd/csrc/host/gpu_iispt_integrator.h"

int main(int argc, char **argv) {
    if (argc != 6) return 2;
    iile::IisptSchedule s(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    for (int i = 0, n = atoi(argv[5]); i < n; ++i) {
        const iile::IisptScheduleTask t = s.Next();
        printf("%d %d %d %d %d %d %d\n", t.x0, t.y0, t.x1, t.y1, t.tilesize, t.pass, t.taskNumber);
    }
    return 0;
}
