/* C-only client of libvgh.so (no Python, no torch): vgh_create from a .vghpack, one vgh_ctx_detect on a raw u8 batch, results
 * written as raw little-endian arrays.  tests/test_gpu_parity.py::test_c_only_create_and_detect compiles this with hipcc, runs it
 * and compares the bytes with the Python path (same pack, same images).
 *   c_abi_smoke <pack> <images.u8> <B> <conf> <out_prefix> */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vgh.h"

#define CHECK_HIP(e)                                                              \
    do {                                                                          \
        hipError_t _e = (e);                                                      \
        if (_e != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d hip error %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return 2;                                                             \
        }                                                                         \
    } while (0)

static int dump(const char* prefix, const char* name, const void* dev, size_t bytes) {
    char path[1024];
    void* host = malloc(bytes ? bytes : 1);
    FILE* f;
    if (hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    snprintf(path, sizeof(path), "%s.%s", prefix, name);
    f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(host, 1, bytes, f);
    fclose(f);
    free(host);
    return 0;
}

int main(int argc, char** argv) {
    vgh_config cfg;
    vgh_ctx* ctx = NULL;
    vgh_ctx_info info;
    vgh_detect_out out;
    int B, rc, cap;
    float conf;
    size_t img_bytes;
    unsigned char* himg;
    void *dimg, *unpad;
    FILE* f;
    float* hun;
    int i;
    if (argc != 6) {
        fprintf(stderr, "usage: %s pack images.u8 B conf out_prefix\n", argv[0]);
        return 1;
    }
    if (vgh_abi_version() != VGH_ABI_VERSION) { /* the header this client was compiled against describes other struct sizes than the library expects */
        fprintf(stderr, "libvgh.so has ABI revision %d, vgh.h %d\n", vgh_abi_version(), VGH_ABI_VERSION);
        return 1;
    }
    B = atoi(argv[3]);
    conf = (float)atof(argv[4]);
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = 0;
    cfg.pack_path = argv[1];
    cfg.max_batch = B;
    /* a wrong path must fail with a message, not crash */
    {
        vgh_config bad = cfg;
        bad.pack_path = "/nonexistent.vghpack";
        if (vgh_create(&bad, &ctx) == VGH_OK || !strstr(vgh_last_error(), "cannot open")) {
            fprintf(stderr, "expected a clean failure for a missing pack, got: %s\n", vgh_last_error());
            return 3;
        }
    }
    rc = vgh_create(&cfg, &ctx);
    if (rc != VGH_OK) {
        fprintf(stderr, "vgh_create failed (%d): %s\n", rc, vgh_last_error());
        return 3;
    }
    if (vgh_ctx_get_info(ctx, &info) != VGH_OK) return 3;
    printf("%s image_size %d anchors %d pre_k %d keep_k %d vertices %d arena_batch %d %.2f GFLOP/image\n", info.variant, info.image_size, info.num_anchors, info.pre_nms_top_k,
           info.keep_top_k, info.num_vertices, info.arena_batch, info.flops_per_image / 1e9);
    img_bytes = (size_t)B * info.image_size * info.image_size * 3;
    himg = (unsigned char*)malloc(img_bytes);
    f = fopen(argv[2], "rb");
    if (!f || fread(himg, 1, img_bytes, f) != img_bytes) {
        fprintf(stderr, "cannot read %zu image bytes from %s\n", img_bytes, argv[2]);
        return 1;
    }
    fclose(f);
    CHECK_HIP(hipMalloc(&dimg, img_bytes));
    CHECK_HIP(hipMemcpy(dimg, himg, img_bytes, hipMemcpyHostToDevice));
    cap = B * info.keep_top_k;
    memset(&out, 0, sizeof(out));
    CHECK_HIP(hipMalloc((void**)&out.boxes_dev, (size_t)cap * 4 * 4));
    CHECK_HIP(hipMalloc((void**)&out.scores_dev, (size_t)cap * 4));
    CHECK_HIP(hipMalloc((void**)&out.flame_dev, (size_t)cap * VGH_NUM_FLAME_PARAMS * 4));
    CHECK_HIP(hipMalloc((void**)&out.counts_dev, (size_t)B * 4));
    CHECK_HIP(hipMalloc((void**)&out.n_heads_dev, 4));
    CHECK_HIP(hipMalloc((void**)&out.head_image_dev, (size_t)cap * 4));
    CHECK_HIP(hipMalloc((void**)&out.rpy_dev, (size_t)cap * 3 * 4));
    CHECK_HIP(hipMalloc((void**)&out.proj_dev, (size_t)cap * info.num_vertices * 3 * 4));
    CHECK_HIP(hipMemset(out.proj_dev, 0, (size_t)cap * info.num_vertices * 3 * 4));
    CHECK_HIP(hipMemset(out.rpy_dev, 0, (size_t)cap * 3 * 4));
    CHECK_HIP(hipMemset(out.head_image_dev, 0, (size_t)cap * 4));
    out.head_capacity = cap;
    hun = (float*)malloc((size_t)B * 3 * 4);
    for (i = 0; i < B; ++i) {
        hun[3 * i] = 3.0f * i;
        hun[3 * i + 1] = 5.0f;
        hun[3 * i + 2] = 1.0f + 0.25f * i;
    }
    CHECK_HIP(hipMalloc(&unpad, (size_t)B * 3 * 4));
    CHECK_HIP(hipMemcpy(unpad, hun, (size_t)B * 3 * 4, hipMemcpyHostToDevice));
    out.unpad_dev = (const float*)unpad;
    rc = vgh_ctx_detect(ctx, dimg, VGH_IMG_U8_NHWC, B, conf, 0.5f, &out, NULL);
    if (rc != VGH_OK) {
        fprintf(stderr, "vgh_ctx_detect failed (%d): %s\n", rc, vgh_ctx_last_error(ctx));
        return 4;
    }
    CHECK_HIP(hipDeviceSynchronize());
    /* a call that must fail leaves its message on the context */
    if (vgh_ctx_detect(ctx, dimg, VGH_IMG_U8_NHWC, B + 1, conf, 0.5f, &out, NULL) == VGH_OK || !vgh_ctx_last_error(ctx)[0]) {
        fprintf(stderr, "expected B > max_batch to fail with a per-context message\n");
        return 4;
    }
    if (dump(argv[5], "counts", out.counts_dev, (size_t)B * 4) || dump(argv[5], "boxes", out.boxes_dev, (size_t)cap * 16) || dump(argv[5], "scores", out.scores_dev, (size_t)cap * 4) ||
        dump(argv[5], "flame", out.flame_dev, (size_t)cap * VGH_NUM_FLAME_PARAMS * 4) || dump(argv[5], "n_heads", out.n_heads_dev, 4) ||
        dump(argv[5], "head_image", out.head_image_dev, (size_t)cap * 4) || dump(argv[5], "rpy", out.rpy_dev, (size_t)cap * 12) ||
        dump(argv[5], "proj", out.proj_dev, (size_t)cap * info.num_vertices * 12))
        return 5;
    vgh_destroy(ctx);
    printf("ok\n");
    return 0;
}
