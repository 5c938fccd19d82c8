// ptrender: the reference's main() without its window (src/main.rs:62-118) — build a scene
// with the host scene API, build the BVHs, call Camera::render on a GPU and save the image
// (PNG as the reference's strat.png when the output name ends in .png, PPM otherwise).
//
//     ptrender [width] [spp] [out.ppm]                     the Cornell scene
//     ptrender object <file.obj> [width] [spp] [out.ppm]   the Object scene on an .obj file
//
// The scene functions are src/scenes/cornell.rs:18-133 and src/scenes/object.rs:20-96 written
// against pt:: — they show what a user of the reference's scene API finds here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "pthost.hpp"

using namespace pt;

struct Cornell {
    static Camera build_camera(uint32_t width, uint32_t spp) {
        return Camera(1.0f, width, 40.0f, Point3(278.0f, 278.0f, -800.0f), Point3(278.0f, 278.0f, 0.0f), 1.0f, 0.0f, spp, 16,
                      Color(0.0f, 0.0f, 0.0f));
    }
    static void build_scene(PrimitiveBuffer &objects, LightBuffer &lights) {
        MaterialRegistry materials;
        materials.create_material("red", Lambertian::make(Color(0.65f, 0.05f, 0.05f)));
        materials.create_material("mirror", Metal::make(Color(0.8f, 0.85f, 0.88f), 0.05f));
        materials.create_material("white", Lambertian::make(Color(0.73f, 0.73f, 0.73f)));
        materials.create_material("green", Lambertian::make(Color(0.12f, 0.45f, 0.15f)));
        materials.create_material("light", Diffuse::make(Color(15.0f, 15.0f, 15.0f)));
        materials.create_material("glass", DisneyGlass::make(Color(1.0f, 1.0f, 1.0f), 0.3f, 0.0f, 1.5f));

        Quad q1(Point3(555, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), materials.get("green"));
        Quad q2(Point3(0, 0, 0), Vector3(0, 555, 0), Vector3(0, 0, 555), materials.get("red"));
        Quad q3(Point3(343, 554, 332), Vector3(-130, 0, 0), Vector3(0, 0, -105), materials.get("light"));
        Quad q4(Point3(0, 0, 0), Vector3(555, 0, 0), Vector3(0, 0, 555), materials.get("white"));
        Quad q5(Point3(555, 555, 555), Vector3(-555, 0, 0), Vector3(0, 0, -555), materials.get("white"));
        Quad q6(Point3(0, 0, 555.5f), Vector3(555, 0, 0), Vector3(0, 555, 0), materials.get("white"));
        objects.add_quad(q1);
        objects.add_quad(q2);
        lights.add_quad(q3);
        objects.add_quad(q3);
        objects.add_quad(q4);
        objects.add_quad(q5);
        objects.add_quad(q6);

        Instance box1 = Instance::from_obj(Quad::new_box(Point3(0, 0, 0), Point3(165, 330, 165), materials.get("mirror")));
        box1.rotate_y(15.0f * (3.14159265358979323846f / 180.0f));
        box1.translate(Vector3(265, 0, 295));
        PrimitiveBuffer sph;
        sph.add_sphere(Sphere(Point3(190, 90, 190), 90.0f, materials.get("glass")));
        objects.add_instance(box1);
        objects.add_instance(Instance::from_obj(std::move(sph)));
    }
};

struct Object {
    static Camera build_camera(uint32_t width, uint32_t spp) {
        return Camera(1.0f, width, 15.0f, Point3(2.0f, 10.0f, 16.0f), Point3(0.0f, 1.0f, 0.0f), 10.0f, 0.0f, spp, 50, Color(0.0f, 0.0f, 0.0f));
    }
    static void build_scene(const std::string &obj_path, PrimitiveBuffer &objects, LightBuffer &lights) {
        MaterialRegistry materials;
        std::string why;
        const std::vector<TriMesh> meshes = TriMesh::load_as_vec(obj_path, &why);
        if (meshes.empty()) std::fprintf(stderr, "Failed to load mesh: %s... Using empty mesh instead...\n", why.c_str());
        TextureRef lite = Solid::make(Color(0.6f, 0.6f, 0.6f)), dark = Solid::make(Color(0.2f, 0.2f, 0.2f));
        TextureRef blue = Solid::make(Color(0.8f, 0.6f, 0.2f));
        TextureRef check = Checkered::make(lite, dark, 13.0f);
        materials.create_material("emit", Diffuse::gray(20.0f));
        materials.create_material("check", Lambertian::new_texture(check));
        materials.create_material("mat1", Lambertian::make(Color(0.2f, 0.2f, 0.2f)));
        materials.create_material("test", Disney::make(blue, 0.5f, 0.1f, 0.0f, 0.01f, 0.2f, 0.1f, 0.8f, 0.3f, 0.1f, 0.2f, 0.5f, 1.1f));
        Sphere light(Point3(-10.0f, 15.0f, 5.0f), 2.0f, materials.get("emit"));
        lights.add_sphere(light);
        objects.add_sphere(light);
        const char *mats[3] = {"check", "mat1", "test"};
        if (meshes.size() > 3) throw std::runtime_error("the Object scene has materials for three models");
        for (size_t i = 0; i < meshes.size(); i++)
            for (Triangle &tri : meshes[i].to_triangles_with_mat(materials.get(mats[i]))) objects.add_triangle(std::move(tri));
    }
};

int main(int argc, char **argv) {
    const bool object = argc > 2 && std::string(argv[1]) == "object";
    const int a0 = object ? 3 : 1;
    const uint32_t width = argc > a0 ? (uint32_t)std::atoi(argv[a0]) : 768;
    const uint32_t spp = argc > a0 + 1 ? (uint32_t)std::atoi(argv[a0 + 1]) : 1024;
    const char *out = argc > a0 + 2 ? argv[a0 + 2] : "strat.ppm";
    try {
        std::fprintf(stderr, "Building scene...\n");
        auto t0 = std::chrono::steady_clock::now();
        Camera camera = object ? Object::build_camera(width, spp) : Cornell::build_camera(width, spp);
        PrimitiveBuffer objects;
        LightBuffer lights;
        if (object) Object::build_scene(argv[2], objects, lights);
        else Cornell::build_scene(objects, lights);
        objects.build_bvh();
        auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "Build time: %.3f ms\nRendering...\n", std::chrono::duration<double, std::milli>(t1 - t0).count());
        RgbImage image;
        camera.render(objects, lights, image, 0, 0,
                      [](uint32_t done, uint32_t total, void *) { std::fprintf(stderr, "\r%u / %u passes", done, total); }, nullptr);
        auto t2 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "\nRender time: %.3f s\n", std::chrono::duration<double>(t2 - t1).count());
        const std::string out_path = out;
        const bool png = out_path.size() >= 4 && out_path.compare(out_path.size() - 4, 4, ".png") == 0;  // the reference writes PNG (src/main.rs:93)
        if (!(png ? image.save_png(out) : image.save_ppm(out))) throw std::runtime_error("cannot write output image");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "ptrender: %s\n", e.what());
        return 1;
    }
    return 0;
}
