// examples/spheres.rs: the random-spheres scene on a flat World.
#include "../host/scenes.hpp"

static const float ASPECT_RATIO = 16.0f / 9.0f;
static const uint32_t SCREEN_WIDTH = 600;
static const uint32_t SCREEN_HEIGHT = (uint32_t)((float)SCREEN_WIDTH / ASPECT_RATIO);
static const uint32_t SAMPLES_PER_PIXEL = 10;
static const uint32_t MAX_DEPTH = 10;

int main() {
  rt::seed(1);
  rt::World world = rt::scenes::random_spheres();
  rt::Camera camera = rt::scenes::weekend_camera(ASPECT_RATIO);
  auto renderer = rt::Renderer::create(SCREEN_WIDTH, SCREEN_HEIGHT, SAMPLES_PER_PIXEL, MAX_DEPTH, rt::Color(0.8f, 0.8f, 0.8f));
  if (!renderer.is_ok()) { std::fprintf(stderr, "%s\n", renderer.error.c_str()); return 1; }
  rt::Status st = renderer.unwrap().render(world, camera);
  if (!st.is_ok()) { std::fprintf(stderr, "%s\n", st.error.c_str()); return 1; }
  st = renderer.unwrap().present();
  if (!st.is_ok()) { std::fprintf(stderr, "%s\n", st.error.c_str()); return 1; }
  return 0;
}
