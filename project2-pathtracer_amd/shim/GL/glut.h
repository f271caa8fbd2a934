/* GL/glut.h -- headless shim: the GLUT calls of /root/reference/src/main.cpp:85-89,181-198,236-240.
 * glutMainLoop() calls the registered display callback for ever (`for(;;) display();`): the process ends when runCuda()
 * calls exit(0) after saving the frame asked for with `frame=` (main.cpp:158-161). */
#ifndef PTMI355_SHIM_GLUT_H
#define PTMI355_SHIM_GLUT_H
#include "glew.h"
#define GLUT_RGBA 0
#define GLUT_DOUBLE 2
#ifdef __cplusplus
extern "C" {
#endif
void glutInit(int *argc, char **argv);
void glutInitDisplayMode(unsigned int mode);
void glutInitWindowSize(int width, int height);
int glutCreateWindow(const char *title);
void glutDisplayFunc(void (*func)(void));
void glutKeyboardFunc(void (*func)(unsigned char key, int x, int y));
void glutMainLoop(void);
void glutSetWindowTitle(const char *title);
void glutPostRedisplay(void);
void glutSwapBuffers(void);
#ifdef __cplusplus
}
#endif
#endif
