// ref_bitmap_harness.cpp — TEST INFRASTRUCTURE, never part of the product.
//
// The reference's framebuffer class (image_bitmap.h:11-75) compiles on its own, so it is compiled here unmodified from
// where it lies under /root/reference (`make -C oracle ref`, output oracle/_ref/ref_bitmap) and driven through a fixed
// script of calls; what it leaves in getRawData() is the fixture tests/golden/ref_bitmap.bin.  Built a second time with
// -DBITMAP_STAND_IN=<header> the same script runs on the stand-in class the drop-in's unit test is compiled against, which
// must produce the same bytes: layout data[(y*W + x)*3 + c], silently dropped out-of-range setPixel (image_bitmap.h:36-49),
// clipped copyImageTo (:67-74), clear (:63-65).
//
//   ref_bitmap <out.bin> [config.json]    out: for each step of the script: int32 width, height, then width*height*3 floats
//
// The reference build also includes config.h (config.h:8-20) and writes its constants, and how the comparisons against the
// double-valued EPS come out for the floats around 1e-4 (the device code and the oracle use float-equivalent forms of them).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#ifdef BITMAP_STAND_IN
#include BITMAP_STAND_IN
#else
#include "config.h"
#include "image_bitmap.h"
#endif

static void dump(FILE *f, const image_bitmap &img)
{
    int32_t head[2] = {img.getWidth(), img.getHeight()};
    size_t n = (size_t)head[0] * head[1] * 3;
    if (fwrite(head, 4, 2, f) != 2 || fwrite(img.getRawData(), 4, n, f) != n) { perror("write"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc != 2 && argc != 3) { fprintf(stderr, "usage: ref_bitmap <out.bin> [config.json]\n"); return 2; }
#ifndef BITMAP_STAND_IN
    if (argc == 3) {
        FILE *c = fopen(argv[2], "w");
        if (!c) { perror(argv[2]); return 2; }
        fprintf(c, "{\"WIDTH\": %d, \"HEIGHT\": %d, \"RENDER_COUNT\": %d, \"AO_RAYS_COUNT\": %d, \"MAX_REFLECTION_DEPTH\": %d,\n"
                   " \"THREAD_POOL_SIZE\": %d, \"SUB_BLOCK_WIDTH\": %d, \"SUB_BLOCK_HEIGHT\": %d, \"EPS\": %.17g, \"sizeof_EPS\": %d,\n",
                WIDTH, HEIGHT, RENDER_COUNT, AO_RAYS_COUNT, MAX_REFLECTION_DEPTH, THREAD_POOL_SIZE, SUB_BLOCK_WIDTH, SUB_BLOCK_HEIGHT,
                (double)EPS, (int)sizeof(EPS));
#ifdef GPU_ACCELERATION
        fprintf(c, " \"GPU_ACCELERATION\": true,");
#else
        fprintf(c, " \"GPU_ACCELERATION\": false,");
#endif
#ifdef ENABLE_AO
        fprintf(c, " \"ENABLE_AO\": true,");
#else
        fprintf(c, " \"ENABLE_AO\": false,");
#endif
#ifdef ENABLE_REFLECTION
        fprintf(c, " \"ENABLE_REFLECTION\": true,");
#else
        fprintf(c, " \"ENABLE_REFLECTION\": false,");
#endif
#ifdef RENDER_PARALLEL
        fprintf(c, " \"RENDER_PARALLEL\": true,\n");
#else
        fprintf(c, " \"RENDER_PARALLEL\": false,\n");
#endif
        // x < EPS, fabs(x) < EPS (ray_tracer.cpp:305,311), x > EPS (:191) for the nine floats around 0.0001f
        float x0 = 0.0001f;
        uint32_t bits; memcpy(&bits, &x0, 4);
        fprintf(c, " \"eps_float_bits\": %u, \"around_eps\": [", bits);
        for (int k = -4; k <= 4; k++) {
            uint32_t b = bits + (uint32_t)k; float x; memcpy(&x, &b, 4);
            fprintf(c, "%s[%d, %d, %d]", k == -4 ? "" : ", ", (int)(x < EPS), (int)(x > EPS), (int)(x <= 0.0001f));
        }
        fprintf(c, "]}\n");
        fclose(c);
    }
#endif
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 2; }
    image_bitmap a(7, 5);
    dump(f, a);                                                     // zero-initialised (image_bitmap.h:19-26)
    for (int y = -2; y < 8; y++)                                    // in and out of range on every side
        for (int x = -3; x < 10; x++) a.setPixel(x, y, 100.0f * y + x, 0.5f * x, -1.0f * y);
    dump(f, a);
    float *p = a.getPixel(2, 3);                                    // getPixel points into the same storage
    p[0] = 7.25f; p[1] = 8.5f; p[2] = 9.75f;
    dump(f, a);
    image_bitmap b(3, 4);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 3; x++) b.setPixel(x, y, 1000.0f + 10.0f * y + x, 2000.0f + x, 3000.0f + y);
    a.copyImageTo(5, 3, b);                                         // clipped at the right and the top edge
    dump(f, a);
    a.copyImageTo(-1, -2, b);                                       // negative offsets: what falls outside is dropped
    dump(f, a);
    a.copyImageTo(1, 0, b);
    dump(f, a);
    a.clear();
    dump(f, a);
    image_bitmap c(1, 1);
    c.setPixel(0, 0, 1.0f, 2.0f, 3.0f);
    c.setPixel(1, 0, 9.0f, 9.0f, 9.0f);
    c.setPixel(0, 1, 9.0f, 9.0f, 9.0f);
    dump(f, c);
    fclose(f);
    return 0;
}
