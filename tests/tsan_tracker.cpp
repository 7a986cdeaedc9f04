// ThreadSanitizer driver for the host v_pref tracker's thread pool (sca_amd/csrc/sca_dubins.hpp: Pool, step_all).
// Test infrastructure (tests/test_sanitizers.py builds it with -fsanitize=thread): a few hundred agents on random poses,
// several steps with the pool, re-sized once; the same run single-threaded must give the same v_pref bit for bit.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "sca_dubins.hpp"

int main() {
    const int n = 192, steps = 6;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-30.0, 30.0), A(-3.0, 3.0);
    auto make = [&](sca_dubins::Tracker &T) {
        T.n = n;
        T.goal.resize(3 * n); T.goal_heading.assign(3 * n, 0.0); T.pref_speed.assign(n, 1.0); T.zaxis.assign(n, 0);
        T.st.assign(n, sca_dubins::AgentTrack());
    };
    sca_dubins::Tracker T1, T4;
    make(T1); make(T4);
    std::vector<double> pos(3 * n), heading(3 * n, 0.0), nb0(n, -1.0), v1(3 * n), v4(3 * n);
    std::vector<float> vel(3 * n, 0.0f);
    std::vector<uint8_t> active(n, 1);
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { pos[3 * i + k] = U(rng); T1.goal[3 * i + k] = T4.goal[3 * i + k] = U(rng); }
        pos[3 * i + 2] = 40.0 + 0.2 * pos[3 * i + 2]; T1.goal[3 * i + 2] = T4.goal[3 * i + 2] = 40.0 + 0.2 * T1.goal[3 * i + 2];
        heading[3 * i] = A(rng); T1.goal_heading[3 * i] = T4.goal_heading[3 * i] = A(rng);
    }
    int bad = 0;
    for (int s = 0; s < steps; s++) {
        sca_dubins::step_all(T1, pos.data(), vel.data(), heading.data(), active.data(), nb0.data(), v1.data(), 1);
        sca_dubins::step_all(T4, pos.data(), vel.data(), heading.data(), active.data(), nb0.data(), v4.data(), s < 3 ? 4 : 3);
        for (int i = 0; i < 3 * n; i++) if (std::memcmp(&v1[i], &v4[i], sizeof(double)) != 0) bad++;
        for (int i = 0; i < n; i++)                                     // move along v_pref, a float32 velocity as the env stores it
            for (int k = 0; k < 3; k++) { vel[3 * i + k] = (float)v1[3 * i + k]; pos[3 * i + k] += 0.1 * (double)vel[3 * i + k]; }
    }
    delete T4.pool;
    std::printf("tsan_tracker: %d agents, %d steps, %d mismatching components\n", n, steps, bad);
    return bad ? 1 : 0;
}
