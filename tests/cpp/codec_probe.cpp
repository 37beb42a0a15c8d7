// codec_probe.cpp — test driver for integration/png_min.hpp and integration/json_min.hpp.  Same command line and
// output layout as oracle/ref_codec_harness.cpp, so a test can compare the two byte for byte:
//   codec_probe png  <in.png>  <out.bin>   uint32 error (0 ok, 1 rejected), width, height, RGBA8 bytes
//   codec_probe json <in.json> <out.bin>   {"floats": [...], "ints": [...]} -> uint32 n, float[n], uint32 m, int32[m], float[m]
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "json_min.hpp"
#include "png_min.hpp"

static void put(FILE *f, const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { perror("write"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: codec_probe png|json <in> <out>\n"); return 2; }
    FILE *out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 2; }
    try {
        if (!strcmp(argv[1], "png")) {
            uint32_t head[3] = {0, 0, 0};
            rt_hip::ImageRGBA8 img;
            try { img = rt_hip::decode_png_file(argv[2]); head[1] = img.width; head[2] = img.height; }
            catch (const std::exception &e) { head[0] = 1; fprintf(stderr, "%s\n", e.what()); }
            put(out, head, sizeof head);
            if (!head[0]) put(out, img.rgba.data(), img.rgba.size());
        } else {
            std::vector<uint8_t> bytes = rt_hip::read_file(argv[2]);
            rt_hip::Json doc = rt_hip::Json::parse(std::string(bytes.begin(), bytes.end()));
            std::vector<float> fl; std::vector<int32_t> iv; std::vector<float> ifl;
            const rt_hip::Json &a = doc.get("floats"), &b = doc.get("ints");
            for (size_t i = 0; i < a.size(); i++) fl.push_back(a.at(i).f32());
            for (size_t i = 0; i < b.size(); i++) { iv.push_back(b.at(i).i32()); ifl.push_back(b.at(i).f32()); }
            uint32_t n = (uint32_t)fl.size();
            put(out, &n, 4); put(out, fl.data(), 4 * fl.size());
            n = (uint32_t)iv.size();
            put(out, &n, 4); put(out, iv.data(), 4 * iv.size()); put(out, ifl.data(), 4 * ifl.size());
        }
    } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); fclose(out); return 1; }
    fclose(out);
    return 0;
}
