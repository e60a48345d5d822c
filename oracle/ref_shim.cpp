// TEST INFRASTRUCTURE.  C-linkage entry points over the reference's own rasteriser so that Python (ctypes) can call it.
// This file contains NO reference code: it only declares the two functions of
//   /root/reference/head_detector/Sim3DR/lib/rasterize.h:88-95  (_rasterize_triangles, _rasterize)
// and forwards to them.  oracle/build_ref.py compiles it together with
//   /root/reference/head_detector/Sim3DR/lib/rasterize_kernel.cpp   (from where it lies; never copied)
// into oracle/_ref/libsim3dr_ref.so -- the *reference itself*, used to pin oracle/raster_oracle.py and the HIP kernel.
void _rasterize(unsigned char* image, float* vertices, int* triangles, float* colors, float* depth_buffer, int ntri, int h, int w, int c, float alpha,
                bool reverse);
void _rasterize_triangles(float* vertices, int* triangles, float* depth_buffer, int* triangle_buffer, float* barycentric_weight, int ntri, int h, int w);

extern "C" {
void ref_rasterize(unsigned char* image, float* vertices, int* triangles, float* colors, float* depth_buffer, int ntri, int h, int w, int c, float alpha,
                   int reverse) {
    _rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha, reverse != 0);
}
void ref_rasterize_triangles(float* vertices, int* triangles, float* depth_buffer, int* triangle_buffer, float* barycentric_weight, int ntri, int h, int w) {
    _rasterize_triangles(vertices, triangles, depth_buffer, triangle_buffer, barycentric_weight, ntri, h, w);
}
}
