// ref_bitmap_harness.cpp — TEST INFRASTRUCTURE, never part of the product.
//
// The reference's framebuffer class (image_bitmap.h:11-75) compiles on its own, so it is compiled here unmodified from
// where it lies under /root/reference (`make -C oracle ref`, output oracle/_ref/ref_bitmap) and driven through a fixed
// script of calls; what it leaves in getRawData() is the fixture tests/golden/ref_bitmap.bin.  Built a second time with
// -DBITMAP_STAND_IN=<header> the same script runs on the stand-in class the drop-in's unit test is compiled against, which
// must produce the same bytes: layout data[(y*W + x)*3 + c], silently dropped out-of-range setPixel (image_bitmap.h:36-49),
// clipped copyImageTo (:67-74), clear (:63-65).
//
//   ref_bitmap <out.bin>     out: for each step of the script: int32 width, height, then width*height*3 floats
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#ifdef BITMAP_STAND_IN
#include BITMAP_STAND_IN
#else
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
    if (argc != 2) { fprintf(stderr, "usage: ref_bitmap <out.bin>\n"); return 2; }
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
