// The reference's four scene programs written against the host mirror
// (examples/spheres.rs, spheres_bvh.rs, cornel_box.rs, final_scene.rs), plus
// the large synthetic variant of the random-spheres scene.  Scene randomness
// comes from rt::rng(); call rt::seed() first.
#pragma once
#include <string>

#include "rt.hpp"

namespace rt {
namespace scenes {

using Matte = Lambertian<SolidTexture>;

// examples/spheres.rs:42-107.  The reference loops `for a in -11..11`; the
// synthetic stress scene widens the grid to [-half, half] inclusive.
inline World random_spheres(int half = 11, bool inclusive = false) {
  World world;
  auto material_ground = Lambertian<CheckerTexture<SolidTexture, SolidTexture>>(
      checker_from_colors(Color(0.2f, 0.3f, 0.1f), Color(0.9f, 0.9f, 0.9f)));
  world.add(Sphere<decltype(material_ground)>(Point3(0.0f, -1000.0f, 0.0f), 1000.0f, material_ground));

  const int end = inclusive ? half + 1 : half;
  for (int a = -half; a < end; ++a) {
    for (int b = -half; b < end; ++b) {
      const float choose_mat = rng().uniform01();
      const float jx = rng().uniform01();
      const float jz = rng().uniform01();
      const Point3 center((float)a + 0.9f * jx, 0.2f, (float)b + 0.9f * jz);
      if ((center - Point3(4.0f, 0.2f, 0.0f)).length() > 0.9f) {
        if (choose_mat < 0.8f) {
          const Matte material_lambertian(SolidTexture::from_color(Color::random(0.0f, 0.5f)));
          const Point3 center2 = center + Vec3(0.0f, rng().gen_range(0.0f, 0.5f), 0.0f);
          world.add(MovingSphere<Matte>(center, center2, 0.0f, 1.0f, 0.2f, material_lambertian));
        } else if (choose_mat < 0.95f) {
          const SolidTexture albedo = SolidTexture::from_color(Color::random(0.5f, 1.0f));
          const Metal<SolidTexture> material_metal(albedo, rng().uniform01());
          world.add(Sphere<Metal<SolidTexture>>(center, 0.2f, material_metal));
        } else {
          world.add(Sphere<Dielectric>(center, 0.2f, Dielectric(1.5f)));
        }
      }
    }
  }
  world.add(Sphere<Dielectric>(Point3(0.0f, 1.0f, 0.0f), 1.0f, Dielectric(1.5f)));
  world.add(Sphere<Matte>(Point3(-4.0f, 1.0f, 0.0f), 1.0f, Matte(SolidTexture::from_color(Color(0.4f, 0.2f, 0.1f)))));
  world.add(Sphere<Metal<SolidTexture>>(Point3(4.0f, 1.0f, 0.0f), 1.0f,
                                        Metal<SolidTexture>(SolidTexture::from_color(Color(0.7f, 0.6f, 0.5f)), 0.0f)));
  return world;
}

// examples/cornel_box.rs:58-105
inline World cornell_box() {
  World world;
  const Matte red(SolidTexture::from_rgb(0.65f, 0.05f, 0.05f));
  const Matte white(SolidTexture::from_rgb(0.73f, 0.73f, 0.73f));
  const Matte green(SolidTexture::from_rgb(0.12f, 0.45f, 0.15f));
  const DiffuseLight<SolidTexture> light(SolidTexture::from_rgb(15.0f, 15.0f, 15.0f));

  world.add(YZRect<Matte>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, green));
  world.add(YZRect<Matte>(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, red));
  world.add(FlipFace<XZRect<DiffuseLight<SolidTexture>>>(
      XZRect<DiffuseLight<SolidTexture>>(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, light)));
  world.add(XZRect<Matte>(0.0f, 555.0f, 0.0f, 555.0f, 0.0f, white));
  world.add(XZRect<Matte>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white));
  world.add(XYRect<Matte>(0.0f, 555.0f, 0.0f, 555.0f, 555.0f, white));
  world.add(Translate<Rotate<Box3d<Matte>>>(
      Rotate<Box3d<Matte>>(Box3d<Matte>(Point3(0.0f, 0.0f, 0.0f), Point3(165.0f, 330.0f, 165.0f), white), 15.0f),
      Vec3(265.0f, 0.0f, 295.0f)));
  world.add(Sphere<Dielectric>(Point3(190.0f, 90.0f, 190.0f), 90.0f, Dielectric(2.0f)));
  return world;
}

// examples/cornel_box.rs:12-26
inline World cornell_lights() {
  World lights;
  const Matte dummy_material(SolidTexture::from_color(Color(0.4f, 0.2f, 0.1f)));
  lights.add(XZRect<Matte>(213.0f, 343.0f, 227.0f, 332.0f, 554.0f, dummy_material));
  lights.add(Sphere<Matte>(Point3(190.0f, 90.0f, 190.0f), 90.0f, dummy_material));
  return lights;
}

// examples/final_scene.rs:48-171.  `earth_path` replaces "textures/earthmap.jpg".
inline Result<World> final_scene(const std::string& earth_path) {
  World boxes;
  const Matte ground(SolidTexture::from_rgb(0.48f, 0.83f, 0.53f));
  const uint32_t boxes_per_side = 20;
  for (uint32_t i = 0; i < boxes_per_side; ++i) {
    for (uint32_t j = 0; j < boxes_per_side; ++j) {
      const float w = 100.0f;
      const float x0 = -1000.0f + (float)i * w;
      const float z0 = -1000.0f + (float)j * w;
      const float y0 = 0.0f;
      const float x1 = x0 + w;
      const float y1 = (float)rng().gen_range_i(1, 101);
      const float z1 = z0 + w;
      boxes.add(Box3d<Matte>(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
    }
  }
  World world;
  world.add(std::move(boxes));

  const DiffuseLight<SolidTexture> light(SolidTexture::from_rgb(7.0f, 7.0f, 7.0f));
  world.add(FlipFace<XZRect<DiffuseLight<SolidTexture>>>(
      XZRect<DiffuseLight<SolidTexture>>(123.0f, 423.0f, 147.0f, 412.0f, 554.0f, light)));

  const Point3 center1(400.0f, 400.0f, 400.0f);
  const Point3 center2 = center1 + Point3(30.0f, 0.0f, 0.0f);
  world.add(MovingSphere<Matte>(center1, center2, 0.0f, 1.0f, 50.0f, Matte(SolidTexture::from_rgb(0.7f, 0.3f, 0.1f))));
  world.add(Sphere<Dielectric>(Point3(260.0f, 150.0f, 45.0f), 50.0f, Dielectric(1.5f)));
  world.add(Sphere<Metal<SolidTexture>>(Point3(0.0f, 150.0f, 145.0f), 50.0f,
                                        Metal<SolidTexture>(SolidTexture::from_rgb(0.8f, 0.8f, 0.8f), 1.0f)));

  using Fog = ConstantMedium<Sphere<Dielectric>, Matte>;
  world.add(Fog(Sphere<Dielectric>(Point3(360.0f, 150.0f, 145.0f), 70.0f, Dielectric(1.5f)), 0.2f,
                Matte(SolidTexture::from_rgb(0.2f, 0.4f, 0.9f))));
  world.add(Fog(Sphere<Dielectric>(Point3(0.0f, 0.0f, 0.0f), 5000.0f, Dielectric(1.5f)), 0.0001f,
                Matte(SolidTexture::from_rgb(1.0f, 1.0f, 1.0f))));

  auto earth = ImageTexture::open(earth_path);
  if (!earth.is_ok()) return Result<World>::err(earth.error);
  world.add(Sphere<Lambertian<ImageTexture>>(Point3(400.0f, 200.0f, 400.0f), 100.0f,
                                             Lambertian<ImageTexture>(earth.unwrap())));

  world.add(Sphere<Lambertian<NoiseTexture>>(Point3(220.0f, 280.0f, 300.0f), 80.0f,
                                             Lambertian<NoiseTexture>(NoiseTexture(0.1f))));

  World boxes2;
  const Matte white(SolidTexture::from_rgb(0.73f, 0.73f, 0.73f));
  for (int i = 0; i < 1000; ++i) boxes2.add(Sphere<Matte>(Point3::random(0.0f, 165.0f), 10.0f, white));
  world.add(Translate<Rotate<World>>(Rotate<World>(std::move(boxes2), 15.0f), Vec3(-100.0f, 270.0f, 395.0f)));
  return Result<World>::ok(std::move(world));
}

// examples/final_scene.rs:13-16
inline XZRect<Matte> final_scene_light() {
  return XZRect<Matte>(123.0f, 423.0f, 147.0f, 412.0f, 554.0f, Matte(SolidTexture::from_color(Color(0.4f, 0.2f, 0.1f))));
}

// examples/spheres.rs:12-28
inline Camera weekend_camera(float aspect_ratio) {
  return Camera(Point3(13.0f, 2.0f, 3.0f), Point3(0.0f, 0.0f, 0.0f), Point3(0.0f, 1.0f, 0.0f), 40.0f, aspect_ratio,
                0.1f, 10.0f, 0.0f, 1.0f);
}
// examples/cornel_box.rs:28-44, final_scene.rs:18-34
inline Camera cornell_camera(float aspect_ratio) {
  return Camera(Point3(478.0f, 278.0f, -600.0f), Point3(278.0f, 278.0f, 0.0f), Point3(0.0f, 1.0f, 0.0f), 40.0f,
                aspect_ratio, 0.0f, 10.0f, 0.0f, 1.0f);
}

}  // namespace scenes
}  // namespace rt
