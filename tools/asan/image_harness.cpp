#include "pathtracer.h"
#include <cstdio>
int main(int argc, char** argv)
{
    long total = 0;
    for (int i = 1; i < argc; i++) { Image im(argv[i]); if (im.data()) total += im.width() * im.height(); }
    std::printf("decoded pixels %ld\n", total);
    return 0;
}
