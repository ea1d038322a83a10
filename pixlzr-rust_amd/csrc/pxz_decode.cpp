// pxz_decode — the reference CLI's `pix_to_image` flow (src/bin/main.rs) on the C++ mirror:
//   .pixlzr -> Pixlzr::open -> to_image(filter) -> raw interleaved pixels
// Raw output instead of a PNG encoder (the `image` crate's encoders are outside the path).
//   pxz_decode <in.pixlzr> <filter 0..4 | file> <out.raw>       prints "<width> <height> <channels> <blocks>"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>

#include "../../include/pixlzr.hpp"

int main(int argc, char **argv)
{
	if (argc != 4) {
		std::fprintf(stderr, "usage: %s in.pixlzr filter|file out.raw\n", argv[0]);
		return 2;
	}
	try {
		const pixlzr::Pixlzr pix = pixlzr::Pixlzr::open(argv[1]);
		// From<Pixlzr> for DynamicImage: the file's own filter, Gaussian when it has none (pixlzr_image.rs:77-81)
		const pixlzr::FilterType f = !std::strcmp(argv[2], "file") ? pix.filter.value_or(pixlzr::FilterType::Gaussian)
		                                                           : (pixlzr::FilterType)std::atoi(argv[2]);
		const pixlzr::Pixlzr::Image img = pix.to_image(f);
		std::ofstream o(argv[3], std::ios::binary);
		o.write(reinterpret_cast<const char *>(img.data.data()), (std::streamsize)img.data.size());
		std::printf("%u %u %u %zu\n", img.width, img.height, img.channels, pix.blocks.size());
		return 0;
	} catch (const std::exception &e) {
		std::fprintf(stderr, "pxz_decode: %s\n", e.what());
		return 1;
	}
}
