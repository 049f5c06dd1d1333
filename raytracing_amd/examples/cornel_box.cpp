// examples/cornel_box.rs: Cornell box with a World of lights for the mixture pdf.
#include "../host/scenes.hpp"

static const float ASPECT_RATIO = 1.0f;
static const uint32_t SCREEN_WIDTH = 1000;
static const uint32_t SCREEN_HEIGHT = (uint32_t)((float)SCREEN_WIDTH / ASPECT_RATIO);
static const uint32_t SAMPLES_PER_PIXEL = 5;
static const uint32_t MAX_DEPTH = 5;

int main() {
  rt::World world = rt::scenes::cornell_box();
  rt::World lights = rt::scenes::cornell_lights();
  rt::Camera camera = rt::scenes::cornell_camera(ASPECT_RATIO);
  auto renderer = rt::Renderer::create(SCREEN_WIDTH, SCREEN_HEIGHT, SAMPLES_PER_PIXEL, MAX_DEPTH, rt::Color(0.0f, 0.0f, 0.0f));
  if (!renderer.is_ok()) { std::fprintf(stderr, "%s\n", renderer.error.c_str()); return 1; }
  rt::Status st = renderer.unwrap().render(world, camera, &lights);
  if (!st.is_ok()) { std::fprintf(stderr, "%s\n", st.error.c_str()); return 1; }
  st = renderer.unwrap().present();
  if (!st.is_ok()) { std::fprintf(stderr, "%s\n", st.error.c_str()); return 1; }
  return 0;
}
