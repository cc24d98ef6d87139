// plan_sim_main.cpp -- stand-alone driver of the plan checker for the sanitizer build (tests/test_plan_host.py):
// the planner (csrc/plan.cpp) and the checker (plan_sim.cpp) under -fsanitize=address,undefined, no Python in the process.
// usage: plan_sim_asan nblk numeric want_grad slack mutate [...]   (five integers per case) -> one line per case
#include <cstdio>
#include <cstdlib>
extern "C" int plan_sim(int nblk, int numeric, int want_grad, int slack, int mutate, double* rep, char* msg, int msglen);
int main(int argc, char** argv) {
    for (int i = 1; i + 4 < argc; i += 5) {
        double rep[8] = {0};
        char msg[512] = {0};
        const int rc = plan_sim(atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2]), atoi(argv[i + 3]), atoi(argv[i + 4]), rep, msg, 512);
        printf("%d %.3e %.3e %.3e %.3e %.0f | %s\n", rc, rep[0], rep[1], rep[2], rep[3], rep[4], msg);
    }
    return 0;
}
