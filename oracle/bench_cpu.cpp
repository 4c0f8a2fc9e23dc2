// oracle/bench_cpu.cpp -- the CPU leg of bench.py: the CPU restatement (oracle/) timed on the host cores of the box,
// pure C++ (no Python in the timed loop), g++ -O3 -ffp-contract=off (the reference is x86-64 -O3 without FMA).
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (see mml_oracle.h); never part of the product path.
//
// One "scan" is what one bench.py step does per scan: getVeloFeature + getHoriFeature (detectFeaturePoints per ring /
// line), RemoveLidarDistortion of the fused cloud, label split + VoxelGrid, ONE association pass against the local map
// (kd-trees built once, outside the timing, like the GPU grid) and `gn_iters` fixed trust-region iterations.
// Three variants over the same scans:
//   single            one thread
//   reference_shaped  the reference's own threading: 6 threads over the Livox lines, Velodyne rings serial
//                     (unionFeatureExtract.cpp:1008-1015, 1228-1230), corner || surf association in two threads
//                     (Estimator.cpp:1271-1297), 6 threads in the solve (:1430)
//   scan_parallel     one scan per core: `--threads` pinned worker processes (what a CPU box can do for THROUGHPUT;
//                     the reference itself never does this)
// Workload file (little endian), written by bench.py:
//   int magic 0x424c4d4d, int n_scans, int n_rings, float pitch0, pitch_step, near, far, int n_lines, float leaf_corner,
//   leaf_surf, int gn_iters, double thres_dist; per scan: int n_velo, float[4 n_velo], int n_livox, 20-byte records,
//   double dR[9], dt[3], x0[6]; then int m_corner, float[3 m], int m_surf, float[3 m].
#include <malloc.h>
#include <pthread.h>
#include <sched.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mml_oracle.h"

namespace {

struct Scan {
    std::vector<float> velo;
    std::vector<mmlo_livox_point> livox;
    double dR[9], dt[3], x0[6];
};
struct Workload {
    int n_rings = 16, n_lines = 6, gn_iters = 10;
    float pitch0 = -15, pitch_step = 2, near_th = 2, far_th = 50, leaf_corner = 0.4f, leaf_surf = 0.2f;
    double thres = 25;
    std::vector<Scan> scans;
    std::vector<float> cmap, smap;
    mmlo_kdtree *tc = nullptr, *ts = nullptr;
};

template <typename T>
bool rd(FILE* f, T* p, size_t n) {
    return n == 0 || fread(p, sizeof(T), n, f) == n;
}

bool load(const char* path, Workload& w) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    int magic = 0, ns = 0;
    bool ok = rd(f, &magic, 1) && magic == 0x424c4d4d && rd(f, &ns, 1) && rd(f, &w.n_rings, 1) && rd(f, &w.pitch0, 1) &&
              rd(f, &w.pitch_step, 1) && rd(f, &w.near_th, 1) && rd(f, &w.far_th, 1) && rd(f, &w.n_lines, 1) &&
              rd(f, &w.leaf_corner, 1) && rd(f, &w.leaf_surf, 1) && rd(f, &w.gn_iters, 1) && rd(f, &w.thres, 1);
    if (ok) w.scans.resize(ns);
    for (int k = 0; ok && k < ns; ++k) {
        Scan& s = w.scans[k];
        int n = 0;
        ok = rd(f, &n, 1);
        if (ok) s.velo.resize(4 * (size_t)n);
        ok = ok && rd(f, s.velo.data(), s.velo.size()) && rd(f, &n, 1);
        if (ok) s.livox.resize(n);
        ok = ok && rd(f, s.livox.data(), s.livox.size()) && rd(f, s.dR, 9) && rd(f, s.dt, 3) && rd(f, s.x0, 6);
    }
    int m = 0;
    ok = ok && rd(f, &m, 1);
    if (ok) w.cmap.resize(3 * (size_t)m);
    ok = ok && rd(f, w.cmap.data(), w.cmap.size()) && rd(f, &m, 1);
    if (ok) w.smap.resize(3 * (size_t)m);
    ok = ok && rd(f, w.smap.data(), w.smap.size());
    fclose(f);
    return ok;
}

// per-thread scratch, sized once
struct Scratch {
    std::vector<float> xyzi, rel, xyz, cfeat, sfeat, csel, ssel;
    std::vector<int> line, label;
    std::vector<mmlo_line_factor> lf;
    std::vector<mmlo_plane_factor> pf;
    void size(size_t n) {
        xyzi.resize(4 * n);
        rel.resize(n);
        xyz.resize(3 * n);
        line.resize(n);
        label.resize(n);
        cfeat.resize(3 * n);
        sfeat.resize(3 * n);
        csel.resize(3 * n);
        ssel.resize(3 * n);
        lf.resize(n);
        pf.resize(n);
    }
};

void run_scan(const Workload& w, const Scan& s, Scratch& t, bool shaped, double* x_out) {
    const int nv_in = (int)s.velo.size() / 4, nl_in = (int)s.livox.size();
    t.size((size_t)nv_in + nl_in + 1);
    int c0, c1;
    // feature node: getVeloFeature, then getHoriFeature (unionCloudHandler calls them one after the other)
    const int nv = mmlo_extract_velo(s.velo.data(), nv_in, w.n_rings, w.pitch0, w.pitch_step, w.near_th, w.far_th, t.xyzi.data(),
                                     t.rel.data(), t.line.data(), t.label.data(), &c0, &c1);
    const int nl = mmlo_extract_livox(s.livox.data(), nl_in, w.n_lines, w.near_th, w.far_th, t.xyzi.data() + 4 * (size_t)nv,
                                      t.rel.data() + nv, t.line.data() + nv, t.label.data() + nv, &c0, &c1);
    const int n = nv + nl;
    // pose node: RemoveLidarDistortion on the merged cloud (unionPoseEstimation.cpp:746-757, 862)
    for (int i = 0; i < n; ++i) {
        t.xyz[3 * i] = t.xyzi[4 * i];
        t.xyz[3 * i + 1] = t.xyzi[4 * i + 1];
        t.xyz[3 * i + 2] = t.xyzi[4 * i + 2];
    }
    mmlo_undistort(t.xyz.data(), t.rel.data(), n, s.dR, s.dt);
    // label split + VoxelGrid (Estimator.cpp:992-1026)
    int ncs = 0, nss = 0;
    for (int i = 0; i < n; ++i) {
        if (t.label[i] == 1) {
            memcpy(&t.csel[3 * (size_t)ncs++], &t.xyz[3 * (size_t)i], 12);
        } else if (t.label[i] == 2) {
            memcpy(&t.ssel[3 * (size_t)nss++], &t.xyz[3 * (size_t)i], 12);
        }
    }
    const int nc = mmlo_voxel_downsample(t.csel.data(), ncs, w.leaf_corner, t.cfeat.data());
    const int nsf = mmlo_voxel_downsample(t.ssel.data(), nss, w.leaf_surf, t.sfeat.data());
    // transformTobeMapped from x0 = [t, phi], extrinsic identity
    double q[4], T[16];
    mmlo_so3_exp(s.x0 + 3, q);
    const double x = q[0], y = q[1], z = q[2], ww = q[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww), 2 * (x * y + z * ww), 1 - 2 * (x * x + z * z),
                         2 * (y * z - x * ww), 2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = s.x0[r];
    }
    T[12] = T[13] = T[14] = 0;
    T[15] = 1;
    int n_line = 0, n_plane = 0;
    if (shaped) {  // threads[0] = processPointToLine, threads[1] = processPointToPlanVec (Estimator.cpp:1271-1297)
        std::thread th([&] {
            n_line = mmlo_associate_lines(t.cfeat.data(), nc, w.cmap.data(), (int)w.cmap.size() / 3, w.tc, T, w.thres, t.lf.data(), nullptr);
        });
        n_plane = mmlo_associate_planes(t.sfeat.data(), nsf, w.smap.data(), (int)w.smap.size() / 3, w.ts, T, w.thres, t.pf.data(), nullptr);
        th.join();
    } else {
        n_line = mmlo_associate_lines(t.cfeat.data(), nc, w.cmap.data(), (int)w.cmap.size() / 3, w.tc, T, w.thres, t.lf.data(), nullptr);
        n_plane = mmlo_associate_planes(t.sfeat.data(), nsf, w.smap.data(), (int)w.smap.size() / 3, w.ts, T, w.thres, t.pf.data(), nullptr);
    }
    const double T_bl[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    mmlo_solve_opts so{w.gn_iters, 1, 0.1 / 1.5e-3, 0.0};
    mmlo_solve_summary sm;
    double xs[6];
    memcpy(xs, s.x0, sizeof(xs));
    mmlo_solve_window(t.lf.data(), &n_line, t.pf.data(), &n_plane, 1, T_bl, &so, xs, &sm, nullptr);
    if (x_out) memcpy(x_out, xs, sizeof(xs));
}

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void pin(int core) {
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(core, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

std::string cpu_model() {
    FILE* f = fopen("/proc/cpuinfo", "r");
    char line[512];
    std::string m = "unknown";
    while (f && fgets(line, sizeof(line), f))
        if (!strncmp(line, "model name", 10)) {
            const char* c = strchr(line, ':');
            if (c) {
                m = c + 2;
                while (!m.empty() && (m.back() == '\n' || m.back() == ' ')) m.pop_back();
            }
            break;
        }
    if (f) fclose(f);
    for (auto& ch : m)
        if (ch == '"' || ch == '\\') ch = ' ';
    return m;
}

}  // namespace

// ---- one scan per core: one PROCESS per core (fork after the kd-trees exist; they are shared copy-on-write).  Threads of
// one process queue on the address-space lock whenever the allocator maps or trims memory, which at a few hundred
// workers costs most of the machine; separate address spaces do not.  Returns scans/s summed over the workers.
static double run_processes(const Workload& w, const std::vector<int>& cores, int workers, double seconds, long* total) {
    const int ns = (int)w.scans.size();
    fflush(stdout);
    std::vector<int> fds(workers, -1);
    std::vector<pid_t> pids(workers, -1);
    for (int i = 0; i < workers; ++i) {
        int pf[2];
        if (pipe(pf) != 0) break;
        const pid_t pid = fork();
        if (pid == 0) {
            close(pf[0]);
            pin(cores[i % cores.size()]);
            mmlo_set_threading(1, 1);
            Scratch sc;
            long done = 0;
            const double t1 = now();
            long k = i;  // workers start on different scans
            while (now() - t1 < seconds) {
                run_scan(w, w.scans[k % ns], sc, false, nullptr);
                ++k;
                ++done;
            }
            const double el = now() - t1;
            double rec[2] = {(double)done, el};
            if (write(pf[1], rec, sizeof(rec)) != (ssize_t)sizeof(rec)) _exit(1);
            _exit(0);
        }
        close(pf[1]);
        fds[i] = pf[0];
        pids[i] = pid;
    }
    double rate_sum = 0;
    *total = 0;
    for (int i = 0; i < workers; ++i) {
        if (fds[i] < 0) continue;
        double rec[2] = {0, 1};
        if (read(fds[i], rec, sizeof(rec)) == (ssize_t)sizeof(rec) && rec[1] > 0) {
            *total += (long)rec[0];
            rate_sum += rec[0] / rec[1];  // every worker ran for the same time: the rates add
        }
        close(fds[i]);
        if (pids[i] > 0) waitpid(pids[i], nullptr, 0);
    }
    return rate_sum;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: bench_cpu workload.bin [--single-seconds S] [--shaped-seconds S] [--parallel-seconds S] [--threads N]\n");
        return 2;
    }
    double t_single = 6, t_shaped = 6, t_par = 8;
    int threads = (int)std::thread::hardware_concurrency();
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--single-seconds")) t_single = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--shaped-seconds")) t_shaped = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--parallel-seconds")) t_par = atof(argv[i + 1]);
        if (!strcmp(argv[i], "--threads")) threads = atoi(argv[i + 1]);
    }
    if (threads < 1) threads = 1;
    // keep the per-scan work buffers (a few MB each) inside the malloc arenas: with the default thresholds every scan
    // mmaps / munmaps them, and a few hundred threads then queue on the process's address-space lock instead of computing
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    Workload w;
    if (!load(argv[1], w) || w.scans.empty()) {
        fprintf(stderr, "bench_cpu: cannot read %s\n", argv[1]);
        return 2;
    }
    const double tb0 = now();
    w.tc = mmlo_kdtree_build(w.cmap.data(), (int)w.cmap.size() / 3);
    w.ts = mmlo_kdtree_build(w.smap.data(), (int)w.smap.size() / 3);
    const double t_tree = now() - tb0;
    const int ns = (int)w.scans.size();
    // the cores this process may use (the box may hand us a subset)
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    sched_getaffinity(0, sizeof(allowed), &allowed);
    std::vector<int> cores;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed)) cores.push_back(c);
    if (cores.empty()) cores.push_back(0);
    if (threads > (int)cores.size()) threads = (int)cores.size();
    // a container may see every CPU of the host and still be held to a CPU-time quota (cgroup v2 cpu.max, v1 cfs quota):
    // more workers than the quota only adds throttling, so the process-parallel variant uses ceil(quota) of them
    double quota = 0;
    {
        FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
        char a[64] = {0};
        long period = 0;
        if (f && fscanf(f, "%63s %ld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        if (f) fclose(f);
        if (quota <= 0) {
            long q = -1, per = 0;
            FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
            FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (fq && fp && fscanf(fq, "%ld", &q) == 1 && fscanf(fp, "%ld", &per) == 1 && q > 0 && per > 0) quota = (double)q / (double)per;
            if (fq) fclose(fq);
            if (fp) fclose(fp);
        }
    }
    if (quota > 0 && threads > (int)ceil(quota)) threads = (int)ceil(quota);

    std::vector<double> x_first(6 * (size_t)ns, 0.0);
    // ---- single thread (pinned) ----
    double single_rate = 0;
    int single_n = 0;
    {
        pin(cores[0]);
        mmlo_set_threading(1, 1);
        Scratch sc;
        run_scan(w, w.scans[0], sc, false, nullptr);  // warm-up
        const double t0 = now();
        while (single_n < ns || now() - t0 < t_single) {
            run_scan(w, w.scans[single_n % ns], sc, false, single_n < ns ? &x_first[6 * (size_t)single_n] : nullptr);
            ++single_n;
        }
        single_rate = single_n / (now() - t0);
    }
    // ---- reference-shaped threading: up to 6 cores busy for one scan at a time ----
    double shaped_rate = 0, shaped_pose_diff = 0;
    int shaped_n = 0;
    const int shaped_cores = threads < 6 ? threads : 6;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int i = 0; i < shaped_cores; ++i) CPU_SET(cores[i], &set);
        pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // workers inherit the mask
        mmlo_set_threading(shaped_cores, shaped_cores);
        Scratch sc;
        double xs[6];
        run_scan(w, w.scans[0], sc, true, xs);
        for (int k = 0; k < 6; ++k) shaped_pose_diff = fmax(shaped_pose_diff, fabs(xs[k] - x_first[k]));
        const double t0 = now();
        while (now() - t0 < t_shaped) {
            run_scan(w, w.scans[shaped_n % ns], sc, true, nullptr);
            ++shaped_n;
        }
        shaped_rate = shaped_n / (now() - t0);
        mmlo_set_threading(1, 1);
    }
    // a container may see more CPUs than its CPU-time quota pays for: measured both ways, the better one is reported
    double par_rate = 0;
    long par_n = 0;
    int par_workers = threads;
    {
        par_rate = run_processes(w, cores, threads, t_par, &par_n);
        if (threads < (int)cores.size()) {
            long n2 = 0;
            const double r2 = run_processes(w, cores, (int)cores.size(), t_par, &n2);
            if (r2 > par_rate) {
                par_rate = r2;
                par_n = n2;
                par_workers = (int)cores.size();
            }
        }
    }
    printf("{\"cpu_model\": \"%s\", \"host_cores\": %d, \"cgroup_cpu_quota\": %.2f, \"kdtree_build_s\": %.4f, \"distinct_scans\": %d, "
           "\"single\": {\"scans_per_s\": %.4f, \"scans\": %d, \"cores\": 1}, "
           "\"reference_shaped\": {\"scans_per_s\": %.4f, \"scans\": %d, \"cores\": %d, \"pose_diff_vs_single\": %.3g}, "
           "\"scan_parallel\": {\"scans_per_s\": %.4f, \"scans\": %ld, \"cores\": %d}, \"poses\": [",
           cpu_model().c_str(), (int)cores.size(), quota, t_tree, ns, single_rate, single_n, shaped_rate, shaped_n, shaped_cores, shaped_pose_diff,
           par_rate, par_n, par_workers);
    for (int k = 0; k < ns; ++k)
        printf("%s[%.17g, %.17g, %.17g, %.17g, %.17g, %.17g]", k ? ", " : "", x_first[6 * k], x_first[6 * k + 1], x_first[6 * k + 2],
               x_first[6 * k + 3], x_first[6 * k + 4], x_first[6 * k + 5]);
    printf("]}\n");
    mmlo_kdtree_free(w.tc);
    mmlo_kdtree_free(w.ts);
    return 0;
}
