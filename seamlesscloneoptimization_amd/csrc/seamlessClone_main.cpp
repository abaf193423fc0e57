// seamlessClone_main -- native CLI with the reference's argv (seamlessClone-CUDA/seamlessClone_main.cu:69-94):
//
//     seamlessClone_main src.yml dst.yml mask.yml centerX centerY gpu [out.bmp [solver [dump_dir]]]
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

// One OpenCV-FileStorage matrix, node "data" (what compare/vs.py reads): 8-bit (dt u) or float32 (dt f) values.
template <typename T>
static bool write_yml(const std::string &path, const char *name, const T *v, int rows, int cols)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    const bool is_float = sizeof(T) == sizeof(float);
    fprintf(f, "%%YAML:1.0\n---\nmat_name: %s\ndata: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: %s\n", name, rows, cols, is_float ? "f" : "u");
    std::string cur = "   data: [ ";
    char tok[48];
    const size_t n = (size_t)rows * cols;
    for (size_t i = 0; i < n; ++i) {
        if (is_float) snprintf(tok, sizeof(tok), "%.9g%s", (double)v[i], i + 1 < n ? ", " : " ]");      // 9 significant digits: a float32 round trip
        else snprintf(tok, sizeof(tok), "%d%s", (int)v[i], i + 1 < n ? ", " : " ]");
        if (cur.size() + strlen(tok) > 78) {
            while (!cur.empty() && cur.back() == ' ') cur.pop_back();
            fprintf(f, "%s\n", cur.c_str());
            cur = "       ";
        }
        cur += tok;
    }
    fprintf(f, "%s\n", cur.c_str());
    return fclose(f) == 0;
}

// The reference's SCDEBUG intermediates (seamlessClone_imp.cpp:2110-2117) through the stage-level hooks: DIR/ucMask0.yml (the eroded
// ROI mask) and DIR/g{0,1,2}.yml -- the right-hand side with the Dirichlet ring folded in (seamlessClone_imp.cpp:1983-2009), planes in
// the reference's R, G, B order (:374-376) -- i.e. what compare/vs.py:81-86 diffs against OpenCV's mod_diff{2,1,0}.yml.
static bool dump_rhs(void *inst, const std::string &dir, const Mat8 &patch, const Mat8 &dest, const Mat8 &mask, int cx, int cy)
{
    int geo[6];
    if (sc_hip_mask_stage(inst, mask.data.data(), mask.cols, mask.rows, mask.step(), cx, cy, geo, nullptr, 0) != SC_OK) return false;
    const int W = geo[2], H = geo[3];
    std::vector<uint8_t> M((size_t)W * H);
    if (sc_hip_mask_stage(inst, mask.data.data(), mask.cols, mask.rows, mask.step(), cx, cy, geo, M.data(), M.size()) != SC_OK) return false;
    const size_t plane = (size_t)W * H;
    std::vector<float> B(3 * plane), lap(3 * plane);
    if (sc_hip_build_rhs(inst, patch.data.data(), patch.cols, patch.rows, patch.step(), dest.data.data(), dest.cols, dest.rows, dest.step(),
                         mask.data.data(), mask.cols, mask.rows, mask.step(), cx, cy, geo, B.data(), lap.data(), plane) != SC_OK) return false;
    if (!write_yml(dir + "/ucMask0.yml", "ucMask0", M.data(), H, W)) return false;
    const int w = W - 2, h = H - 2;
    std::vector<float> g((size_t)w * h);
    for (int ch = 0; ch < 3; ++ch) {             // file g<ch> = our plane 2 - ch (ours follow the interleaved B, G, R)
        const float *L = lap.data() + (size_t)(2 - ch) * plane, *D = B.data() + (size_t)(2 - ch) * plane;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float v = L[(size_t)(y + 1) * W + x + 1];
                if (x == 0) v -= D[(size_t)(y + 1) * W];
                if (y == 0) v -= D[x + 1];
                if (x == w - 1) v -= D[(size_t)(y + 1) * W + W - 1];
                if (y == h - 1) v -= D[(size_t)(H - 1) * W + x + 1];
                g[(size_t)y * w + x] = v;
            }
        char name[8];
        snprintf(name, sizeof(name), "g%d", ch);
        if (!write_yml(dir + "/" + name + ".yml", name, g.data(), h, w)) return false;
    }
    printf("wrote %s/ucMask0.yml, g0.yml, g1.yml, g2.yml (%dx%d)\n", dir.c_str(), w, h);
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
    if (argc < 7 || argc > 10) {
        fprintf(stderr, "usage: %s src.yml dst.yml mask.yml centerX centerY gpu [out.bmp [auto|mg|exact|dst|fft [dump_dir]]]\n"
                        "  auto  (default) direct FFT solve in double up to 720 unknowns per side (and elongated ROIs), mg above (SC_METHOD_AUTO)\n"
                        "  mg    multigrid + float-table correction: the reference's arithmetic\n"
                        "  exact multigrid, exact solution of the 5-point system (SC_FLAG_EXACT_TABLES)\n"
                        "  dst   the reference's direct DST solve on the fp64 matrix cores (SC_METHOD_DST)\n"
                        "  fft   the reference's default back-end: FFT-based direct solve, float32 (SC_METHOD_FFT)\n"
                        "  dump_dir (an existing directory): the reference's SCDEBUG intermediates ucMask0.yml, g0.yml, g1.yml, g2.yml\n", argv[0]);
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
    const std::string solver = argc >= 9 ? argv[8] : "auto";
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
    if (argc >= 10 && !dump_rhs(inst, argv[9], patch, dest, mask, cx, cy)) {
        fprintf(stderr, "cannot write the intermediates to %s: %s\n", argv[9], sc_hip_last_error(inst));
        my_seamlessclone_api_imp_destroy(inst);
        return EXIT_FAILURE;
    }
    my_seamlessclone_api_imp_destroy(inst);
    return EXIT_SUCCESS;
}
