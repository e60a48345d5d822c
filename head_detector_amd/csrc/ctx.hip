// Whole-pipeline context behind the C ABI: vgh_create(config{pack path}) builds the network, the FLAME layer and the fused
// detector from ONE .vghpack file (written by `python -m head_detector_amd.pack`), so a C / C++ / cgo / JNI caller needs no
// Python at run time.  SURVEY.md 8(b): "vgh_create(const vgh_config*, vgh_ctx**) (device id, variant, image_size, max_batch,
// weight-pack path, FLAME-pack path) / vgh_destroy".
//
// Replaces HeadDetector.__init__ (head_detector/detector.py:19-30: hub download + torch.jit.load + FLAMELayer()) and, through
// vgh_ctx_detect, HeadDetector._process + the device arithmetic of _parse_predictions for a whole batch (detector.py:54-90).
#include <stdarg.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "vgh_internal.h"

namespace {

#pragma pack(push, 1)
struct PackHeader {
    char magic[8];  // "VGHPACK\0"
    uint32_t version, header_bytes;
    char variant[32];
    int32_t image_size, precision, n_bufs, n_ops, n_levels, shape_c, expr_c, has_flame;
    int32_t tune_batch, reserved;
    double flops_per_image;
    int64_t n_weights, n_biases;
    int32_t V, NB, NJ, F;
};
#pragma pack(pop)
static_assert(sizeof(PackHeader) == 8 + 8 + 32 + 8 * 4 + 8 + 8 + 16 + 16, "pack header layout");

struct Level {
    int32_t buf, h, w, pitch, stride;
};

bool read_exact(FILE* f, void* dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, f) == bytes; }

}  // namespace

struct vgh_ctx {
    vgh_net* net = nullptr;
    vgh_flame* flame = nullptr;
    vgh_detector* det = nullptr;
    vgh_ctx_info info{};
    char err[1024] = "";
};

static int ctx_fail(vgh_ctx* c, int rc) {
    if (c && rc != VGH_OK) snprintf(c->err, sizeof(c->err), "%s", vgh_last_error());
    return rc;
}

extern "C" {

int vgh_create(const vgh_config* cfg, vgh_ctx** out) {
    VGH_REQUIRE(cfg && out && cfg->pack_path, "vgh_create: null argument");
    VGH_REQUIRE(cfg->max_batch >= 1, "vgh_create: max_batch must be >= 1");
    FILE* f = fopen(cfg->pack_path, "rb");
    VGH_REQUIRE(f, "vgh_create: cannot open pack file %s", cfg->pack_path);
    PackHeader h;
    std::vector<vgh_buf_desc> bufs;
    std::vector<vgh_op_desc> ops;
    std::vector<char> names;
    std::vector<Level> levels;
    std::vector<float> w, b, v_template, shapedirs, posedirs, jreg, lbsw;
    std::vector<int32_t> parents;
    bool ok = read_exact(f, &h, sizeof(h)) && memcmp(h.magic, "VGHPACK", 8) == 0;
    if (ok && h.version != 3) {
        fclose(f);
        VGH_REQUIRE(false, "vgh_create: %s is pack version %u, this library reads version 3 (vgh_buf_desc.scale, r05)", cfg->pack_path, h.version);
    }
    ok = ok && h.n_bufs > 0 && h.n_ops > 0 && h.n_levels > 0 && h.n_levels <= VGH_MAX_LEVELS && h.n_weights > 0 && h.n_biases > 0 && h.header_bytes >= sizeof(h) &&
         fseek(f, h.header_bytes, SEEK_SET) == 0;
    if (ok) {
        // the header's counts size every allocation below: bound them by what the file can actually hold before trusting them
        // (a corrupt pack must fail with an error code, not with std::bad_alloc thrown across the C ABI)
        struct stat st;
        ok = fstat(fileno(f), &st) == 0;
        const int64_t fsz = ok ? (int64_t)st.st_size : 0;
        ok = ok && h.n_bufs < (1 << 20) && h.n_ops < (1 << 20) && h.n_weights < fsz / 4 + 1 && h.n_biases < fsz / 4 + 1 && h.V >= 0 && h.NB >= 0 && h.NJ >= 0 &&
             h.V < (1 << 24) && h.NB < (1 << 16) && h.NJ < (1 << 12);
        if (ok) {
            int64_t need = (int64_t)h.header_bytes + (int64_t)h.n_bufs * sizeof(vgh_buf_desc) + (int64_t)h.n_ops * (sizeof(vgh_op_desc) + 32) + (int64_t)h.n_levels * sizeof(Level) +
                           4 * (h.n_weights + h.n_biases);
            if (h.has_flame) need += 4 * ((int64_t)h.V * 3 + (int64_t)h.V * 3 * h.NB + (int64_t)(h.NJ > 0 ? h.NJ - 1 : 0) * 27 * h.V + 2 * (int64_t)h.NJ * h.V + h.NJ);
            ok = need <= fsz;
        }
    }
    if (ok) {
        bufs.resize(h.n_bufs);
        ops.resize(h.n_ops);
        names.resize((size_t)h.n_ops * 32);
        levels.resize(h.n_levels);
        w.resize(h.n_weights);
        b.resize(h.n_biases);
        ok = read_exact(f, bufs.data(), bufs.size() * sizeof(vgh_buf_desc)) && read_exact(f, ops.data(), ops.size() * sizeof(vgh_op_desc)) &&
             read_exact(f, names.data(), names.size()) && read_exact(f, levels.data(), levels.size() * sizeof(Level)) && read_exact(f, w.data(), w.size() * 4) &&
             read_exact(f, b.data(), b.size() * 4);
    }
    if (ok && h.has_flame) {
        ok = h.V > 0 && h.NB > 0 && h.NJ > 1;
        if (ok) {
            v_template.resize((size_t)h.V * 3);
            shapedirs.resize((size_t)h.V * 3 * h.NB);
            posedirs.resize((size_t)(h.NJ - 1) * 9 * 3 * h.V);
            jreg.resize((size_t)h.NJ * h.V);
            parents.resize(h.NJ);
            lbsw.resize((size_t)h.V * h.NJ);
            ok = read_exact(f, v_template.data(), v_template.size() * 4) && read_exact(f, shapedirs.data(), shapedirs.size() * 4) &&
                 read_exact(f, posedirs.data(), posedirs.size() * 4) && read_exact(f, jreg.data(), jreg.size() * 4) && read_exact(f, parents.data(), parents.size() * 4) &&
                 read_exact(f, lbsw.data(), lbsw.size() * 4);
        }
    }
    fclose(f);
    VGH_REQUIRE(ok, "vgh_create: %s is not a readable .vghpack (bad magic, truncated or inconsistent header)", cfg->pack_path);

    vgh_ctx* c = new vgh_ctx();
    const int pre_k = cfg->pre_nms_top_k > 0 ? cfg->pre_nms_top_k : 1000, keep_k = cfg->keep_top_k > 0 ? cfg->keep_top_k : 100;
    // activation arena: every tensor below 2 GiB (32-bit loader offsets) -> larger batches run in arena-sized chunks
    int64_t per_image = 1;
    for (const vgh_buf_desc& bd : bufs) {
        const int64_t by = (int64_t)bd.h * bd.w * bd.pitch * vgh_fmt_bytes(bd.is_f32);
        if (by > per_image) per_image = by;
    }
    int arena = (int)(((1ll << 31) - 1) / per_image);
    if (arena > cfg->max_batch) arena = cfg->max_batch;
    if (arena < 1) arena = 1;
    int rc = vgh_net_create(cfg->device, h.image_size, arena, bufs.data(), h.n_bufs, ops.data(), h.n_ops, w.data(), h.n_weights, b.data(), h.n_biases, &c->net);
    // per-op tile choices travel as NAMES (indices may differ between library builds); unknown names fall back to the heuristic
    for (int i = 0; rc == VGH_OK && i < h.n_ops; ++i) {
        const char* nm = names.data() + (size_t)i * 32;
        if (ops[i].kind != VGH_OP_CONV || !nm[0]) continue;
        const bool split = h.precision == VGH_FMT_BF16X2 || h.precision == VGH_FMT_F16X2 || h.precision == VGH_FMT_F16;  // the parity modes (and single-plane fp16) index their own tile table
        const int ncfg = split ? vgh_conv_split_num_cfgs() : vgh_conv_num_cfgs();
        for (int k = 0; k < ncfg; ++k)
            if (strncmp(split ? vgh_conv_split_cfg_name(k) : vgh_conv_cfg_name(k), nm, 31) == 0) {
                rc = vgh_net_set_cfg(c->net, i, k);
                break;
            }
    }
    int A = 0;
    for (const Level& lv : levels) A += lv.h * lv.w;
    if (rc == VGH_OK && h.has_flame) {
        const int max_heads = cfg->max_heads > 0 ? cfg->max_heads : cfg->max_batch * keep_k;
        rc = vgh_flame_create(cfg->device, h.V, h.NB, h.NJ, v_template.data(), shapedirs.data(), posedirs.data(), jreg.data(), parents.data(), lbsw.data(), max_heads, &c->flame);
    }
    if (rc == VGH_OK) {
        vgh_detect_cfg dc;
        memset(&dc, 0, sizeof(dc));
        dc.n_levels = h.n_levels;
        for (int i = 0; i < h.n_levels; ++i) {
            dc.level_buf[i] = levels[i].buf;
            dc.level_h[i] = levels[i].h;
            dc.level_w[i] = levels[i].w;
            dc.level_pitch[i] = levels[i].pitch;
            dc.level_stride[i] = levels[i].stride;
        }
        dc.shape_live = h.shape_c;
        dc.expr_live = h.expr_c;
        dc.pre_k = pre_k < A ? pre_k : A;
        dc.keep_k = keep_k;
        dc.max_batch = cfg->max_batch;
        rc = vgh_detector_create(c->net, c->flame, &dc, &c->det);
    }
    if (rc == VGH_OK && cfg->batch_split > 1) rc = vgh_net_set_split(c->net, cfg->batch_split);
    if (rc == VGH_OK && cfg->overlap) rc = vgh_detector_set_overlap(c->det, 1);
    if (rc != VGH_OK) {
        vgh_destroy(c);
        return rc;
    }
    memset(&c->info, 0, sizeof(c->info));
    snprintf(c->info.variant, sizeof(c->info.variant), "%.31s", h.variant);
    c->info.image_size = h.image_size;
    c->info.max_batch = cfg->max_batch;
    c->info.arena_batch = arena;
    c->info.num_anchors = A;
    c->info.pre_nms_top_k = pre_k < A ? pre_k : A;
    c->info.keep_top_k = keep_k;
    c->info.num_vertices = h.has_flame ? h.V : 0;
    c->info.shape_live = h.shape_c;
    c->info.expr_live = h.expr_c;
    c->info.precision = h.precision;
    c->info.flops_per_image = h.flops_per_image;
    *out = c;
    return VGH_OK;
}

void vgh_destroy(vgh_ctx* c) {
    if (!c) return;
    if (c->det) vgh_detector_destroy(c->det);
    if (c->flame) vgh_flame_destroy(c->flame);
    if (c->net) vgh_net_destroy(c->net);
    delete c;
}

const char* vgh_ctx_last_error(const vgh_ctx* c) { return c ? c->err : vgh_last_error(); }

int vgh_ctx_get_info(const vgh_ctx* c, vgh_ctx_info* info) {
    VGH_REQUIRE(c && info, "vgh_ctx_get_info: null argument");
    *info = c->info;
    return VGH_OK;
}

int vgh_ctx_detect(vgh_ctx* c, const void* images_dev, int image_fmt, int B, float conf_thr, float iou_thr, vgh_detect_out* out, void* stream) {
    VGH_REQUIRE(c, "vgh_ctx_detect: null context");
    return ctx_fail(c, vgh_detect(c->det, images_dev, image_fmt, B, conf_thr, iou_thr, out, stream));
}

int vgh_ctx_join(vgh_ctx* c, void* stream) {
    VGH_REQUIRE(c, "vgh_ctx_join: null context");
    return ctx_fail(c, vgh_detector_join(c->det, stream));
}

vgh_net* vgh_ctx_net(vgh_ctx* c) { return c ? c->net : nullptr; }
vgh_flame* vgh_ctx_flame(vgh_ctx* c) { return c ? c->flame : nullptr; }
vgh_detector* vgh_ctx_detector(vgh_ctx* c) { return c ? c->det : nullptr; }

}  // extern "C"
