// TEST INFRASTRUCTURE ONLY (see oracle/README.md).  A command-line driver over the reference's own `Encoder` class
// (/root/reference/turing/Encoder.h:42-60), compiled by oracle/Makefile together with the reference's encoder sources
// *where they lie*.  It exists because turing/encode.cpp -- the reference's own driver -- includes a header its CMake
// build generates (turing/git-describe.h, turing/CMakeLists.txt:11-13), and a stand-in for generated code is not
// allowed; `Encoder` itself needs nothing generated.
//
// The driver is linked twice (oracle/Makefile): against the reference's havoc objects (`turing_ref_havoc`) and against
// turingcodec_amd/libhavoc_classic.so (`turing_ref_classic`) -- the same encoder, the primitive tables being the only
// difference.  tests/test_reference_encoder.py asserts that both write the same stream.
//
// What it takes from the reference is interface only: the option names / types / defaults `Encoder` reads from its
// boost::program_options::variables_map (turing/encode.cpp:61-233), and the way an input frame becomes a PictureWrap
// (turing/encode.cpp:341-449, progressive input only).  Usage: same options as `turing encode`, `-o FILE` for the stream.
#include "Encoder.h"
#include "Picture.h"
#include "Speed.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace po = boost::program_options;

// --speed takes a preset name (the reference defines this extractor next to its option table, turing/encode.cpp:46-58)
std::istream &operator>>(std::istream &is, Speed::Type &speed)
{
    std::string name;
    is >> name;
    bool known = false;
#define X(preset) if (name == #preset) { speed = Speed::preset; known = true; }
    ENCODER_SPEED_PRESETS_XMACRO
#undef X
    if (!known) throw po::invalid_option_value(name);
    return is;
}

// part of the C API the reference implements in encode.cpp (turing/turing.h:73, turing/encode.cpp:505-508); the encoder writes it into a
// user-data SEI message (turing/TaskEncodeOutput.cpp:117), so the driver has to answer with the reference's version string
extern "C" const char *turing_version(void) { return "1.1"; }

namespace {

struct IntOpt { const char *name; int def; };
struct BoolOpt { const char *name; bool def; };

void describe(po::options_description &all, std::string &input, std::string &output)
{
    // integers with a default
    static const IntOpt ints[] = {{"bit-depth", 8}, {"atc-sei", -1}, {"qp", 26}, {"aq-depth", 3}, {"aq-range", 6}, {"dqp-depth", -1},
                                  {"max-gop-n", 250}, {"max-gop-m", 8}, {"segment", -1}, {"ctu", 64}, {"min-cu", 8}, {"max-num-merge-cand", 5},
                                  {"threads", 0}, {"concurrent-frames", 4}, {"asm", 1}, {"verbosity", 1}, {"internal-bit-depth", 8}};
    for (const IntOpt &o : ints) all.add_options()(o.name, po::value<int>()->default_value(o.def), "");
    // integers / strings that are absent unless given
    for (const char *n : {"hash", "bitrate"}) all.add_options()(n, po::value<int>(), "");
    for (const char *n : {"dump-pictures", "dump-frames", "mastering-display-info", "sar", "display-window", "overscan", "video-format", "range",
                          "colourprim", "transfer-characteristics", "colour-matrix", "chroma-loc"})
        all.add_options()(n, po::value<std::string>(), "");
    // switches; the tool switches come in pairs (--x / --no-x) that Encoder::booleanSwitchSetting resolves against the speed preset
    static const BoolOpt bools[] = {{"aq", false}, {"shot-change", false}, {"field-coding", false}, {"frame-doubling", false}, {"wpp", true},
                                    {"repeat-headers", false}, {"deblock", true}, {"sao", false}, {"strong-intra-smoothing", false}, {"rqt", false},
                                    {"amp", false}, {"smp", false}, {"rdoq", false}, {"sdh", false}, {"tskip", false}, {"fdm", false}, {"fdam", false},
                                    {"ecu", false}, {"esd", false}, {"cfm", false}, {"met", false}, {"aps", false}, {"rcudepth", false},
                                    {"sao-slow-mode", false}, {"no-parallel-processing", false}, {"force-16", false}};
    for (const BoolOpt &o : bools) all.add_options()(o.name, po::bool_switch()->default_value(o.def), "");
    for (const char *n : {"no-rqt", "no-strong-intra-smoothing", "no-wpp", "no-deblock", "no-sao", "no-rect", "no-amp", "no-smp", "no-fdm", "no-fdam",
                          "no-ecu", "no-esd", "no-cfm", "no-met", "no-sao-slow-mode", "no-rdoq", "no-rcudepth", "no-sdh", "no-tskip", "no-aps"})
        all.add_options()(n, po::bool_switch(), "");
    all.add_options()("input-res", po::value<std::string>()->required(), "")("seek", po::value<size_t>(), "")("frames", po::value<size_t>()->required(), "")(
        "frame-rate", po::value<double>()->required(), "")("output-file,o", po::value<std::string>(&output), "")(
        "speed", po::value<Speed::Type>()->default_value(Speed::slow), "")("psnr", "")("profiler", "")("input-file", po::value<std::string>(&input), "");
}

// one planar 4:2:0 frame of the input file -> the encoder's picture type (8-bit input widened by << 2 when the encoder runs 16-bit samples
// internally, as the reference's driver does)
template <typename Sample>
std::shared_ptr<PictureWrapper> wrap(const Encoder &enc, const std::vector<uint8_t> &frame, int inputBytes, int64_t pts)
{
    auto picture = std::make_shared<PictureWrap<Sample>>(enc.pictureWidth, enc.pictureHeight, 1, 0, 0, 32);
    picture->sampleSize = 8 * sizeof(Sample);
    picture->fieldTB = 0;
    const uint8_t *p = frame.data();
    for (int c = 0; c < 3; ++c)
    {
        const int w = enc.frameWidth >> (c ? 1 : 0), h = enc.frameHeight >> (c ? 1 : 0);
        for (int y = 0; y < h; ++y, p += size_t(w) * inputBytes)
            for (int x = 0; x < (*picture)[c].width; ++x)
                (*picture)[c](x, y) = inputBytes == 2 ? Sample(reinterpret_cast<const uint16_t *>(p)[x]) : Sample(sizeof(Sample) == 2 ? p[x] << 2 : p[x]);
    }
    picture->pts = pts;
    return picture;
}

} // namespace

int main(int argc, const char *argv[])
{
    std::string input, output;
    po::variables_map vm;
    try
    {
        po::options_description all;
        describe(all, input, output);
        po::positional_options_description positional;
        positional.add("input-file", 1);
        po::store(po::command_line_parser(argc, argv).options(all).positional(positional).run(), vm);
        po::notify(vm);
        if (input.empty() || output.empty()) throw std::runtime_error("input file and -o FILE are required");

        Encoder encoder(vm);
        const bool wide = vm["bit-depth"].as<int>() > 8 || vm["internal-bit-depth"].as<int>() > 8;
        const int inputBytes = vm["bit-depth"].as<int>() > 8 ? 2 : 1;
        const size_t frameBytes = size_t(inputBytes) * encoder.frameWidth * encoder.frameHeight * 3 / 2;
        std::ifstream in(input.c_str(), std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + input);
        if (vm.count("seek")) in.seekg(std::streamoff(vm["seek"].as<size_t>() * frameBytes));
        std::ofstream out(output.c_str(), std::ios::binary);
        if (!out) throw std::runtime_error("cannot open " + output);

        std::vector<uint8_t> frame(frameBytes), stream;
        Encoder::PictureMetadata metadata;
        const size_t frames = vm["frames"].as<size_t>();
        for (size_t i = 0; i < frames; ++i)
        {
            in.read(reinterpret_cast<char *>(frame.data()), std::streamsize(frameBytes));
            if (size_t(in.gcount()) != frameBytes) throw std::runtime_error("input file is shorter than --frames");
            stream.clear();
            auto picture = wide ? wrap<uint16_t>(encoder, frame, inputBytes, int64_t(i)) : wrap<uint8_t>(encoder, frame, inputBytes, int64_t(i));
            if (encoder.encodePicture(picture, stream, metadata) && !stream.empty()) out.write(reinterpret_cast<const char *>(stream.data()), std::streamsize(stream.size()));
        }
        for (;;)   // flush: an empty picture until the encoder has nothing more
        {
            stream.clear();
            if (!encoder.encodePicture(nullptr, stream, metadata)) break;
            if (!stream.empty()) out.write(reinterpret_cast<const char *>(stream.data()), std::streamsize(stream.size()));
        }
    }
    catch (std::exception &e)
    {
        std::fprintf(stderr, "%s: %s\n", argv[0], e.what());
        return 1;
    }
    return 0;
}
