// seamlessClone_main -- native CLI with the reference's argv (seamlessClone-CUDA/seamlessClone_main.cu:69-94):
//
//     seamlessClone_main src.yml dst.yml mask.yml centerX centerY gpu [out.bmp]
//
// Reads the three OpenCV-FileStorage yml matrices (node "data", seamlessClone_imp.cu:226-237)
// without OpenCV, runs one warm-up clone and one timed clone through the C ABI (the reference's
// protocol, seamlessClone_imp.cu:303-344), prints the reference's timing line and optionally
// writes the blended image as a 24-bit bottom-up BMP (format of seamlessClone_imp.cu:68-190).
// A plain C++ host: no HIP headers needed, links only libseamlessclone_hip.so.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "../../include/seamlessclone_hip.h"

struct Mat8 {
    int rows = 0, cols = 0, ch = 0;
    std::vector<uint8_t> data;
    int step() const { return cols * ch; }
};

static bool read_yml(const char *path, Mat8 &m)
{
    std::ifstream f(path);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return false; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string t = ss.str();
    size_t p = t.find("!!opencv-matrix");
    if (p == std::string::npos) { fprintf(stderr, "%s: no opencv-matrix\n", path); return false; }
    auto num_after = [&](const char *key) -> int {
        size_t q = t.find(key, p);
        return q == std::string::npos ? -1 : atoi(t.c_str() + q + strlen(key));
    };
    m.rows = num_after("rows:"); m.cols = num_after("cols:");
    size_t q = t.find("dt:", p);
    if (q == std::string::npos || m.rows <= 0 || m.cols <= 0) return false;
    q += 3;
    while (q < t.size() && (t[q] == ' ' || t[q] == '"')) ++q;
    m.ch = 1;
    if (t[q] >= '0' && t[q] <= '9') { m.ch = atoi(t.c_str() + q); while (t[q] >= '0' && t[q] <= '9') ++q; }
    if (t[q] != 'u') { fprintf(stderr, "%s: only 8-bit unsigned matrices are supported\n", path); return false; }
    q = t.find('[', t.find("data:", q));
    if (q == std::string::npos) return false;
    m.data.resize((size_t)m.rows * m.cols * m.ch);
    const char *s = t.c_str() + q + 1;
    for (size_t i = 0; i < m.data.size(); ++i) {
        while (*s && (*s < '0' || *s > '9')) ++s;
        if (!*s) { fprintf(stderr, "%s: truncated data\n", path); return false; }
        m.data[i] = (uint8_t)strtol(s, const_cast<char **>(&s), 10);
    }
    return true;
}

static bool write_bmp(const char *path, const Mat8 &m)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const int pad = (4 - (3 * m.cols) % 4) % 4, row = 3 * m.cols + pad;
    const uint32_t size = 54 + (uint32_t)row * m.rows;
    uint8_t h[54] = { 'B', 'M' };
    auto put32 = [&](int o, uint32_t v) { memcpy(h + o, &v, 4); };
    auto put16 = [&](int o, uint16_t v) { memcpy(h + o, &v, 2); };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, m.cols); put32(22, m.rows);
    put16(26, 1); put16(28, 24); put32(34, (uint32_t)row * m.rows); put32(38, 2835); put32(42, 2835);
    fwrite(h, 1, 54, f);
    const uint8_t zeros[3] = { 0, 0, 0 };
    for (int y = m.rows - 1; y >= 0; --y) {
        fwrite(m.data.data() + (size_t)y * m.step(), 1, 3 * m.cols, f);
        fwrite(zeros, 1, pad, f);
    }
    fclose(f);
    return true;
}

int main(int argc, const char *argv[])
{
    printf("argc: %d\n", argc);
    for (int i = 0; i < argc; ++i) printf("argv[%d]: %s\n", i, argv[i]);
    if (argc < 7 || argc > 9) {
        fprintf(stderr, "usage: %s src.yml dst.yml mask.yml centerX centerY gpu [out.bmp [auto|mg|exact|dst|fft]]\n"
                        "  auto  (default) direct FFT solve in double up to 720 unknowns per side (and elongated ROIs), mg above (SC_METHOD_AUTO)\n"
                        "  mg    multigrid + float-table correction: the reference's arithmetic\n"
                        "  exact multigrid, exact solution of the 5-point system (SC_FLAG_EXACT_TABLES)\n"
                        "  dst   the reference's direct DST solve on the fp64 matrix cores (SC_METHOD_DST)\n"
                        "  fft   the reference's default back-end: FFT-based direct solve, float32 (SC_METHOD_FFT)\n", argv[0]);
        return EXIT_FAILURE;
    }
    Mat8 patch, dest, mask;
    if (!read_yml(argv[1], patch) || !read_yml(argv[2], dest) || !read_yml(argv[3], mask)) return EXIT_FAILURE;
    printf("mat shape: %d, %d, %d\n", patch.cols, patch.rows, patch.ch);
    printf("mat shape: %d, %d, %d\n", dest.cols, dest.rows, dest.ch);
    printf("mat shape: %d, %d, %d\n", mask.cols, mask.rows, mask.ch);
    if (patch.ch != 3 || dest.ch != 3 || mask.ch != 1) { fprintf(stderr, "need 3u, 3u, u matrices\n"); return EXIT_FAILURE; }
    const int cx = atoi(argv[4]), cy = atoi(argv[5]), gpu = atoi(argv[6]);
    void *inst = my_seamlessclone_api_imp_create_instance(gpu);
    if (!inst) return EXIT_FAILURE;
    const std::string solver = argc == 9 ? argv[8] : "auto";
    if (solver != "auto") {
        sc_solver_opts o;
        sc_hip_get_solver(inst, &o);
        if (solver == "mg") o.method = SC_METHOD_MULTIGRID;
        else if (solver == "exact") { o.method = SC_METHOD_MULTIGRID; o.flags |= SC_FLAG_EXACT_TABLES; }
        else if (solver == "dst") o.method = SC_METHOD_DST;
        else if (solver == "fft") o.method = SC_METHOD_FFT;
        else { fprintf(stderr, "unknown solver '%s'\n", solver.c_str()); my_seamlessclone_api_imp_destroy(inst); return EXIT_FAILURE; }
        sc_hip_set_solver(inst, &o);
    }
    Mat8 warm = dest, out = dest;
    int rc = my_seamlessclone_api_imp_run(inst, patch.data.data(), patch.cols, patch.rows, patch.step(), warm.data.data(),
                                          warm.cols, warm.rows, warm.step(), mask.data.data(), mask.cols, mask.rows,
                                          mask.step(), cx, cy, gpu, false);                        // warm up (silent)
    if (rc == SC_OK || rc == SC_ERR_NOT_CONVERGED)
        rc = my_seamlessclone_api_imp_run(inst, patch.data.data(), patch.cols, patch.rows, patch.step(), out.data.data(),
                                          out.cols, out.rows, out.step(), mask.data.data(), mask.cols, mask.rows,
                                          mask.step(), cx, cy, gpu, true);
    if (rc != SC_OK && rc != SC_ERR_NOT_CONVERGED) {
        fprintf(stderr, "seamlessClone failed (%d): %s\n", rc, sc_hip_last_error(inst));
        my_seamlessclone_api_imp_destroy(inst);
        return EXIT_FAILURE;
    }
    sc_run_info info;
    sc_hip_get_info(inst, &info);
    // (the timed call above ran with bSync = true: the library itself printed the reference's two lines,
    //  "Compute stage performance time= ..." and "total device memory used: ...", seamlessClone_imp.cu:345-348)
    printf("device stages: %.3f msec (ROI %dx%d); transfers: H2D %.3f msec, D2H %.3f msec; solver %s%s: %d cycle(s)\n", info.ms_device_total, info.W, info.H, info.ms_h2d, info.ms_d2h, solver.c_str(),
           solver == "auto" ? (info.method == SC_METHOD_FFT ? " -> fft (double)" : " -> mg") : "", info.sweeps);
    if (argc >= 8 && !write_bmp(argv[7], out)) fprintf(stderr, "cannot write %s\n", argv[7]);
    my_seamlessclone_api_imp_destroy(inst);
    return EXIT_SUCCESS;
}
