// oracle/taichi_shim/taichi.h — TEST INFRASTRUCTURE.  Stand-in for the single-header "taichi.h" that
// /root/reference/mls-mpm88.cpp:3 includes (distributed separately from the repository: absent from /root/reference).
// Only what advance() (mls-mpm88.cpp:16-69) and the file's globals need: the fixed-size vectors / matrices, svd and
// polar_decomp of the shim (taichi/common/util.h — see its header for what that pins and what it cannot), and an inert
// GUI so that the file's main() (never called: ref_mpm88_driver.cpp renames it) still compiles.
#pragma once
#include <taichi/common/util.h>

namespace taichi {
struct Canvas {
  struct Shape {
    Shape &radius(real) { return *this; }
    Shape &color(int) { return *this; }
    Shape &close() { return *this; }
  };
  void clear(int) {}
  template <typename... A> Shape rect(A...) { return Shape(); }
  template <typename... A> Shape circle(A...) { return Shape(); }
};
struct GUI {
  Canvas canvas;
  GUI(const char *, int, int) {}
  Canvas &get_canvas() { return canvas; }
  void update() {}
};
}  // namespace taichi
