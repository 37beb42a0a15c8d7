// ref_codec_harness.cpp — TEST INFRASTRUCTURE, never part of the product.
//
// Drives the two third-party pieces of the reference that compile here unmodified and without any
// stand-in: its PNG decoder (lib/lodepng.cpp, LodePNG 20161127) exactly as tex_image::createImage calls
// it (tex_image.h:25-31), and its JSON library (lib/json.h, nlohmann/json 2.1.1) with the conversions
// loadScene applies to numbers (scene.cpp:20-22 stream parse; :53-57 `float x = i[0]`; :97-99,112-114
// `int x = i["center"][0]`).  Built by `make -C oracle ref` from the sources where they lie under
// /root/reference into oracle/_ref/; its outputs become the fixtures of tests/golden/ref_codec/.
//
//   ref_codec png  <in.png>  <out.bin>   out: uint32 error, width, height, then width*height*4 RGBA8 bytes
//   ref_codec json <in.json> <out.bin>   in: {"floats": [...], "ints": [...]}
//                                        out: uint32 n_floats, float[n_floats], uint32 n_ints, int32[n_ints], float[n_ints]
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "lib/json.h"
#include "lib/lodepng.h"

static void put(FILE *f, const void *p, size_t n) { if (n && fwrite(p, 1, n, f) != n) { perror("write"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: ref_codec png|json <in> <out>\n"); return 2; }
    FILE *out = fopen(argv[3], "wb");
    if (!out) { perror(argv[3]); return 2; }
    if (!strcmp(argv[1], "png")) {
        std::vector<unsigned char> data;
        unsigned w = 0, h = 0;
        unsigned err = lodepng::decode(data, w, h, std::string(argv[2]));      // the call of tex_image.h:27
        uint32_t head[3] = {err, w, h};
        put(out, head, sizeof head);
        if (!err) put(out, data.data(), data.size());
    } else if (!strcmp(argv[1], "json")) {
        std::ifstream is(argv[2]);
        nlohmann::json doc;
        is >> doc;                                                             // scene.cpp:20-22
        std::vector<float> fl;
        for (nlohmann::json &v : doc["floats"]) { float x = v; fl.push_back(x); }          // scene.cpp:54
        std::vector<int32_t> iv; std::vector<float> ifl;
        for (nlohmann::json &v : doc["ints"]) { int x = v; iv.push_back(x); float y = v; ifl.push_back(y); }   // scene.cpp:97
        uint32_t n = (uint32_t)fl.size();
        put(out, &n, 4); put(out, fl.data(), 4 * fl.size());
        n = (uint32_t)iv.size();
        put(out, &n, 4); put(out, iv.data(), 4 * iv.size()); put(out, ifl.data(), 4 * ifl.size());
    } else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
    fclose(out);
    return 0;
}
